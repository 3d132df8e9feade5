# two ranks on ONE GPU over gloo: the launcher, the fp16 / bf16 exchange and its report (RCCL needs two GPUs)
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
export SEUNET_DIST_BACKEND=gloo
for dt in fp16 bf16; do
  timeout -k 10 400 python3 bench.py --gpus 2 --steps 5 --warmup 3 --batch 1 --size 64 --dtype $dt --no-cpu-baseline --no-secondary > gpurun_out/r04/dp2_$dt.json 2> gpurun_out/r04/dp2_$dt.err || { tail -20 gpurun_out/r04/dp2_$dt.err; exit 1; }
  python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/r04/dp2_$dt.json') if l.startswith('{')][-1]); print('$dt', d['n_gpus'], round(d['ms_per_step'],2), d['config']['dist_backend'], d['config'].get('grad_allreduce_ms'), d['config']['final_loss'])"
done
SEUNET_DDP_SERIAL=1 timeout -k 10 400 python3 bench.py --gpus 2 --steps 5 --warmup 3 --batch 1 --size 64 --dtype fp16 --no-cpu-baseline --no-secondary > gpurun_out/r04/dp2_fp16_serial.json 2> gpurun_out/r04/dp2_fp16_serial.err
python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/r04/dp2_fp16_serial.json') if l.startswith('{')][-1]); print('fp16 serial', d['n_gpus'], round(d['ms_per_step'],2), d['config'].get('grad_allreduce_ms'))"
unset SEUNET_DIST_BACKEND
timeout -k 10 600 python3 -m pytest tests/test_net_gpu.py -q -x -k "train_mode or ragged or width_mult_2" > gpurun_out/r04/dp_tests.log 2>&1 || { tail -40 gpurun_out/r04/dp_tests.log; exit 1; }
tail -2 gpurun_out/r04/dp_tests.log
