set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
timeout -k 10 900 python3 -m pytest tests/test_net_gpu.py -q -x -s -k "three_classes or golden_fp32 or train_mode or bitwise_reproducible or inference_forward or captured" > gpurun_out/r04/classes.log 2>&1 || { tail -50 gpurun_out/r04/classes.log; exit 1; }
grep -E "n_classes|passed|failed" gpurun_out/r04/classes.log | tail
