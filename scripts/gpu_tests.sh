# GPU test driver used with gpurun (scripts/ travels with the snapshot, gpurun_out/ does not)
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/ops.log 2>&1
rc=$?
rc2=0
echo "ops rc=$rc"; tail -3 gpurun_out/ops.log
if [ $rc -le 1 ]; then
  timeout -k 10 900 python -m pytest tests/test_net_gpu.py -m gpu -q -s -p no:cacheprovider > gpurun_out/net.log 2>&1
  rc2=$?
  echo "net rc=$rc2"; grep -E "HIP-vs-f64|gradient rel-L2|bf16 logits|passed|failed|FAILED" gpurun_out/net.log | tail -20
else
  echo "ops run crashed or timed out (rc=$rc): net tests skipped"
fi
# exit status = the worse of the two runs (a crash / timeout / GPU fault is a failure, not a skip)
[ $rc -ge $rc2 ] && exit $rc
exit $rc2
