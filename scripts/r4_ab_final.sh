set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
for L in ec3 dc6 ec2; do
  for tag in base new base new; do
    if [ $tag = base ]; then export SEUNET_LIB=$PWD/se-unet-airseg_amd/libseunet_hip_base.so; else unset SEUNET_LIB; fi
    echo -n "$L $tag  "; REPS=10 WHICH=fwd,dgrad timeout -k 10 120 python3 scripts/bench_conv.py $L 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/.*| STREAM fwd/STREAM fwd/'
  done
done
unset SEUNET_LIB
for tag in base new base new; do
  if [ $tag = base ]; then export SEUNET_LIB=$PWD/se-unet-airseg_amd/libseunet_hip_base.so; else unset SEUNET_LIB; fi
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-secondary --config none > gpurun_out/r04/ab4_$tag.json 2> gpurun_out/r04/ab4_$tag.err
  python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/r04/ab4_$tag.json') if l.startswith('{')][-1]); c=d['class_ms_per_step']; print('$tag', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), 'conv_fwd', c['conv_fwd'], 'dgrad', c['dgrad'], 'wgrad', c['wgrad'])"
done
