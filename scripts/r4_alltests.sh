# round 4: whole GPU suite + a short bench of the same build (one gpurun call)
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
timeout -k 10 1000 python3 -m pytest tests -m gpu -q -rs --durations=15 > gpurun_out/r04/alltests.log 2>&1 || { tail -60 gpurun_out/r04/alltests.log; exit 1; }
tail -25 gpurun_out/r04/alltests.log
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04/bench_quick.json 2> gpurun_out/r04/bench_quick.err
python3 -c "
import json; d=json.load(open('gpurun_out/r04/bench_quick.json')); print(d['ms_per_step'], d['median_ms_per_step'], d['class_ms_per_step'])"
