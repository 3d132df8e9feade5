# same-box A/B of the marching weight gradient inside the training step
mkdir -p gpurun_out
for tag in tiled march; do
  if [ $tag = tiled ]; then export SEUNET_NO_WGRAD_MARCH=1; else unset SEUNET_NO_WGRAD_MARCH; fi
  timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --dump-kernels gpurun_out/kernels_wm_$tag.tsv > gpurun_out/bench_wm_$tag.log 2>&1
  echo "bench $tag rc=$?"
done
python - <<'PY'
import json
for tag in ("tiled", "march"):
    l=[x for x in open('gpurun_out/bench_wm_%s.log' % tag) if x.startswith('{')]
    if l:
        d=json.loads(l[-1]); print("RESULT %s: %.1f Mvox/s  %.2f ms/step median %.2f  wgrad class %.3f" % (tag, d['value']/1e6, d['ms_per_step'], d['median_ms_per_step'], d['class_ms_per_step']['wgrad']))
    else:
        print(open('gpurun_out/bench_wm_%s.log' % tag).read()[-2000:])
PY
grep -h "wgrad:" gpurun_out/kernels_wm_tiled.tsv | head -30
echo; grep -h "wgrad:" gpurun_out/kernels_wm_march.tsv | head -30
