# round 3: marching weight gradient -- op tests, then single-launch timings against the tiled kernel
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -x -k "weight_gradient" > gpurun_out/r3_wm_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r3_wm_tests.log
for L in dc5 dc3 dc4 ec4 ec5 ec6; do
  WHICH=wgrad timeout -k 10 120 python scripts/bench_conv.py $L 2>&1 | tail -2
done > gpurun_out/r3_wm_bench.log 2>&1
cat gpurun_out/r3_wm_bench.log
