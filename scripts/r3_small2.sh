mkdir -p gpurun_out
timeout -k 10 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --dump-kernels gpurun_out/kernels_small.tsv > gpurun_out/bench_small.log 2>&1
echo "bench rc=$?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/bench_small.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print("RESULT: %.1f Mvox/s  %.2f ms/step median %.2f" % (d['value']/1e6, d['ms_per_step'], d['median_ms_per_step'])); print(d['class_ms_per_step'])
else:
    print(open('gpurun_out/bench_small.log').read()[-2000:])
PY
grep -h "^head\|^loss\|^pool_bwd\|^up_" gpurun_out/kernels_small.tsv
