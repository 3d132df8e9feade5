mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_net_gpu.py -m gpu -q -p no:cacheprovider -x -k "golden or oracle_fp32 or stage or same_choice or three_channel or determin or ragged" > gpurun_out/r3_pool_net.log 2>&1
echo "net rc=$?"; tail -3 gpurun_out/r3_pool_net.log
bash scripts/r3_small2.sh
