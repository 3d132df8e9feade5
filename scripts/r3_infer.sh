mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_net_gpu.py -m gpu -q -p no:cacheprovider -x -s -k "inference_forward or window or 16bit" > gpurun_out/r3_infer_tests.log 2>&1; echo "rc=$?"; grep -E "passed|failed|Error|assert|choices differ" gpurun_out/r3_infer_tests.log | head
timeout -k 10 900 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --config window512 > gpurun_out/bench_w512.log 2>&1; echo "bench rc=$?"
python - <<PY
import json
l=[x for x in open('gpurun_out/bench_w512.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print("step %.2f ms" % d['ms_per_step']); print("window512", d.get('window512')); print("parity", d.get('parity_mode',{}).get('ms_per_step'), "fp16", d.get('fp16_mode',{}).get('ms_per_step'))
else:
    print(open('gpurun_out/bench_w512.log').read()[-3000:])
PY
