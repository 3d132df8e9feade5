mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_net_gpu.py -m gpu -q -p no:cacheprovider -x -k "golden or oracle_fp32 or stage or same_choice or three_channel or determin or ragged or width" > gpurun_out/r3_pool_net.log 2>&1
echo "net rc=$?"; tail -3 gpurun_out/r3_pool_net.log
for tag in kernel defer; do
  if [ $tag = kernel ]; then export SEUNET_NO_POOL_DEFER=1; else unset SEUNET_NO_POOL_DEFER; fi
  timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --dump-kernels gpurun_out/kernels_pool_$tag.tsv > gpurun_out/bench_pool_$tag.log 2>&1
  echo "bench $tag rc=$?"
done
python - <<'PY'
import json
for tag in ("kernel", "defer"):
    l=[x for x in open('gpurun_out/bench_pool_%s.log' % tag) if x.startswith('{')]
    if l:
        d=json.loads(l[-1]); c=d['class_ms_per_step']; print("RESULT %s: %.1f Mvox/s  %.2f ms/step median %.2f  pool_bwd %.3f cat_bwd %.3f in_bwd %.3f" % (tag, d['value']/1e6, d['ms_per_step'], d['median_ms_per_step'], c.get('pool_bwd',0), c['cat_bwd'], c['in_bwd']))
    else:
        print(open('gpurun_out/bench_pool_%s.log' % tag).read()[-2000:])
PY
grep -h "cat_bwd:ec33\|in_bwd:ec33\|cat_bwd:ec63\|in_bwd:ec63" gpurun_out/kernels_pool_kernel.tsv gpurun_out/kernels_pool_defer.tsv
