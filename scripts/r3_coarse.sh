mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_net_gpu.py -m gpu -q -p no:cacheprovider -x -k "golden or oracle_fp32 or same_choice or 16bit or width or ragged or properties" > gpurun_out/r3_coarse_net.log 2>&1
echo "net rc=$?"; tail -3 gpurun_out/r3_coarse_net.log
for tag in old new; do
  if [ $tag = old ]; then export SEUNET_MARCH_NO_COARSE=1; else unset SEUNET_MARCH_NO_COARSE; fi
  timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --dump-kernels gpurun_out/kernels_coarse_$tag.tsv > gpurun_out/bench_coarse_$tag.log 2>&1
  echo "bench $tag rc=$?"
done
python - <<'PY'
import json
for tag in ("old", "new"):
    l=[x for x in open('gpurun_out/bench_coarse_%s.log' % tag) if x.startswith('{')]
    if l:
        d=json.loads(l[-1]); c=d['class_ms_per_step']; print("RESULT %s: %.1f Mvox/s  %.2f ms/step median %.2f  conv_fwd %.3f dgrad %.3f wgrad %.3f" % (tag, d['value']/1e6, d['ms_per_step'], d['median_ms_per_step'], c['conv_fwd'], c['dgrad'], c['wgrad']))
    else:
        print(open('gpurun_out/bench_coarse_%s.log' % tag).read()[-2000:])
PY
