mkdir -p gpurun_out
timeout -k 10 1150 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r3_alltests.log 2>&1
echo "tests rc=$?"; tail -6 gpurun_out/r3_alltests.log
