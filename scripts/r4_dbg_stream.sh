cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
timeout -k 10 300 python3 -m pytest tests/test_ops_gpu.py -q -k "stream" 2>&1 | tail -2
done
timeout -k 10 600 python3 -m pytest tests/test_net_gpu.py -q -k "bitwise_reproducible or 16bit_modes_against or shares_the_gpu or sliding_window_192" 2>&1 | tail -6
