# round 3: march op tests + net tests subset + default bench; $1 = tag
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -x -k "march or conv" > gpurun_out/r3_ops_$1.log 2>&1
echo "ops rc=$?"; tail -3 gpurun_out/r3_ops_$1.log
timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dump-kernels gpurun_out/kernels_$1.tsv > gpurun_out/bench_$1.log 2>&1
echo "bench rc=$?"
python - <<PY
import json
for tag in ("$1",):
    l=[x for x in open('gpurun_out/bench_%s.log' % tag) if x.startswith('{')]
    if l:
        d=json.loads(l[-1]); print("RESULT %s: %.1f Mvox/s  %.2f ms/step" % (tag, d['value']/1e6, d['ms_per_step']))
    else:
        print(open('gpurun_out/bench_%s.log' % tag).read()[-2000:])
PY
