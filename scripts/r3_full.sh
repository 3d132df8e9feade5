# round 3: full GPU test suite + default bench (+ optional A/B against SEUNET_NO_MARCH); $1 = tag
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r3_tests_$1.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r3_tests_$1.log
timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dump-kernels gpurun_out/kernels_$1.tsv > gpurun_out/bench_$1.log 2>&1
echo "bench rc=$?"
SEUNET_NO_MARCH=1 timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_$1_nomarch.log 2>&1
python - <<PY
import json
for tag in ("$1", "$1_nomarch"):
    l=[x for x in open('gpurun_out/bench_%s.log' % tag) if x.startswith('{')]
    if l:
        d=json.loads(l[-1]); print("RESULT %s: %.1f Mvox/s  %.2f ms/step" % (tag, d['value']/1e6, d['ms_per_step']))
    else:
        print(open('gpurun_out/bench_%s.log' % tag).read()[-2000:])
PY
