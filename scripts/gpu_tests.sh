mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/ops1.log 2>&1
rc=$?
echo "ops rc=$rc" | tee -a gpurun_out/ops1.log
tail -5 gpurun_out/ops1.log
if [ $rc -le 1 ]; then
  timeout -k 10 700 python -m pytest tests/test_net_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/net1.log 2>&1
  rc2=$?
  echo "net rc=$rc2" | tee -a gpurun_out/net1.log
  tail -5 gpurun_out/net1.log
fi
exit 0
