mkdir -p gpurun_out
timeout -k 10 1150 python -m pytest tests/test_net_gpu.py -m gpu -q -p no:cacheprovider -x -s -k "vs_oracle or three_channel or width_mult_2 or ragged or 128 or data_parallel or sharded" > gpurun_out/r3_nettests.log 2>&1; echo "rc=$?"; grep -E "same .* flips imposed|raw-input|passed|failed|Error|assert|128\^3|vs 2 ranks" gpurun_out/r3_nettests.log | head -40
