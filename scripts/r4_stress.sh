set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
rm -f gpurun_out/r04/stress.log
for cfg in "fp32 1 32 1500" "bf16 1 32 1500" "fp16 2 32 800" "bf16 1 64 400" "fp32 1 64 200"; do
  timeout -k 10 400 python3 tests/stress_shared_gpu.py $cfg >> gpurun_out/r04/stress.log 2>&1 || true
done
grep -E "repetitions|differ|Error" gpurun_out/r04/stress.log | cut -c1-400 | tail -30
