mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_net_gpu.py -m gpu -q -p no:cacheprovider -x -s -k "vs_oracle or three_channel or width_mult_2 or ragged or 128" > gpurun_out/r3_nettests.log 2>&1; echo "rc=$?"; grep -E "same .* flips imposed|passed|failed|Error|assert|128\^3" gpurun_out/r3_nettests.log | head -40
