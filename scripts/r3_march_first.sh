# round 3: first GPU run of the marching conv: op tests, then the microbenchmark against the tiled kernel
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -x -k "march" > gpurun_out/r3_march_tests.log 2>&1
echo "tests rc=$?"; tail -15 gpurun_out/r3_march_tests.log
WHICH=fwd,dgrad REPS=10 timeout -k 10 600 python scripts/bench_conv.py > gpurun_out/r3_march_bench.log 2>&1
echo "bench rc=$?"; cat gpurun_out/r3_march_bench.log | grep -v amdgpu.ids
