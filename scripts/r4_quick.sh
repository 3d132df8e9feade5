# round 4: op-level epilogue tests + a few network tests + short bench
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
timeout -k 10 600 python3 -m pytest tests/test_ops_gpu.py -q -x -k "${OPS_K:-epilogue or gate or cat}" > gpurun_out/r04/quick_ops.log 2>&1 || { tail -40 gpurun_out/r04/quick_ops.log; exit 1; }
tail -3 gpurun_out/r04/quick_ops.log
timeout -k 10 900 python3 -m pytest tests/test_net_gpu.py -q -x -s -k "${NET_K:-vs_oracle_fp32 or 16bit_modes or reproducible or raw_logit or 16bit_modes_equal or overflow_on_one_rank}" > gpurun_out/r04/quick_net.log 2>&1 || { tail -60 gpurun_out/r04/quick_net.log; exit 1; }
grep -E "rel-L2|raw-input|passed|failed|x[369]3.conv1|128\^3" gpurun_out/r04/quick_net.log | tail -40
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --dump-kernels gpurun_out/r04/quick_groups.tsv > gpurun_out/r04/bench_quick.json 2> gpurun_out/r04/bench_quick.err
python3 -c "
import json; d=json.load(open('gpurun_out/r04/bench_quick.json')); print(d['ms_per_step'], d['median_ms_per_step'], d['class_ms_per_step'])"
