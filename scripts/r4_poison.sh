set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
timeout -k 10 600 python3 -m pytest tests/test_net_gpu.py -q -k "prior_contents" > gpurun_out/r04/poison.log 2>&1 || true
grep -E "^FAILED|^E  |passed|failed" gpurun_out/r04/poison.log | cut -c1-400 | head -40
