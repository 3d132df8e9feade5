mkdir -p gpurun_out
timeout -k 10 900 python tests/lowprec_attribution.py 32 2 gpurun_out/r03_lowprec_attribution_32.md > gpurun_out/attr32.log 2>&1; echo "attr rc=$?"; head -16 gpurun_out/r03_lowprec_attribution_32.md; tail -3 gpurun_out/attr32.log
timeout -k 10 900 python -m pytest tests/test_net_gpu.py -m gpu -q -p no:cacheprovider -x -s -k "128" > gpurun_out/r3_t128.log 2>&1; echo "t128 rc=$?"; grep -E "same .* flips imposed|passed|failed" gpurun_out/r3_t128.log
