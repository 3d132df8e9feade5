set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
timeout -k 10 600 python3 -m pytest tests/test_ops_gpu.py -q -x -k "upsample or resample or up2" > gpurun_out/r04/up_ops.log 2>&1 || { tail -30 gpurun_out/r04/up_ops.log; exit 1; }
tail -1 gpurun_out/r04/up_ops.log
for tag in base new base new; do
  if [ $tag = base ]; then export SEUNET_LIB=$PWD/se-unet-airseg_amd/libseunet_hip_base.so; else unset SEUNET_LIB; fi
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-secondary --config none > gpurun_out/r04/ab5_$tag.json 2> gpurun_out/r04/ab5_$tag.err
  python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/r04/ab5_$tag.json') if l.startswith('{')][-1]); c=d['class_ms_per_step']; print('$tag', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), 'up_fwd', c['up_fwd'], 'up_bwd', c['up_bwd'], 'conv_fwd', c['conv_fwd'], 'wgrad', c['wgrad'], 'dgrad', c['dgrad'])"
done
