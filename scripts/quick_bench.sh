# quick correctness (conv ops) + bench with kernel table; $1 = tag
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "conv" 2>&1 | tail -3
timeout -k 10 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --dump-kernels gpurun_out/kernels_$1.tsv > gpurun_out/bench_$1.log 2>&1
python - <<PY
import json
l=[x for x in open('gpurun_out/bench_$1.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print("RESULT $1: %.1f Mvox/s  %.2f ms/step"%(d['value']/1e6, d['ms_per_step']))
else:
    print(open('gpurun_out/bench_$1.log').read()[-2000:])
PY
