set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
for i in 1 2; do
timeout -k 10 600 python3 -m pytest tests/test_net_gpu.py -q -s -k "16bit_modes_against or raw_logit or bitwise_reproducible or 16bit_modes_equal" > gpurun_out/r04/fp16dp_$i.log 2>&1 || true
grep -E "worst gradient|passed|failed|Error" gpurun_out/r04/fp16dp_$i.log || true
done
