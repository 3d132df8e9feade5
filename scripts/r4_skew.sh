set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
for sk in all dc5 all dc5; do
  if [ $sk = dc5 ]; then export SEUNET_MATES_DC5_ONLY=1; else unset SEUNET_MATES_DC5_ONLY; fi
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-secondary --config none > gpurun_out/r04/mates_$sk.json 2> gpurun_out/r04/mates_$sk.err
  python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/r04/mates_$sk.json') if l.startswith('{')][-1]); c=d['class_ms_per_step']; k={x['kernel']:round(x['avg_ms'],4) for x in d['kernels']}; print('mates $sk', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), 'fwd', c['conv_fwd'], 'wgrad', c['wgrad'], 'dgrad', c['dgrad'], k.get('conv_fwd:dc3'), k.get('conv_fwd:dc5'), k.get('wgrad:dc3'))"
done
