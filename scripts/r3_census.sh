mkdir -p gpurun_out
timeout -k 10 900 python tests/flip_census.py 32 2 gpurun_out/r03_flip_census_32.md > gpurun_out/census32.log 2>&1; echo "census32 rc=$?"; head -24 gpurun_out/r03_flip_census_32.md; tail -3 gpurun_out/census32.log
