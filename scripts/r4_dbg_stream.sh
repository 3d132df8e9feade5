cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
  timeout -k 10 300 python3 -m pytest tests/test_ops_gpu.py -q -k "stream" 2>&1 | tail -3
done
