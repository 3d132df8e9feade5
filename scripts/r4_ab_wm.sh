set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
timeout -k 10 600 python3 -m pytest tests/test_ops_gpu.py -q -x -k "wgrad or march" > gpurun_out/r04/wm_ops.log 2>&1 || { tail -30 gpurun_out/r04/wm_ops.log; exit 1; }
tail -1 gpurun_out/r04/wm_ops.log
timeout -k 10 900 python3 -m pytest tests/test_net_gpu.py -q -x -k "16bit_modes_against or full_size_properties or bitwise_reproducible or prior_contents or config4 or same_choice" > gpurun_out/r04/wm_net.log 2>&1 || { tail -30 gpurun_out/r04/wm_net.log; exit 1; }
tail -1 gpurun_out/r04/wm_net.log
for L in dc5 dc3 dc4 ec5 dc1; do
  for tag in base new base new; do
    if [ $tag = base ]; then export SEUNET_LIB=$PWD/se-unet-airseg_amd/libseunet_hip_base.so; else unset SEUNET_LIB; fi
    echo -n "$L $tag  "; REPS=10 WHICH=wgrad timeout -k 10 120 python3 scripts/bench_conv.py $L 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/.*| MARCH wgrad/MARCH wgrad/'
  done
done
unset SEUNET_LIB
for tag in base new base new; do
  if [ $tag = base ]; then export SEUNET_LIB=$PWD/se-unet-airseg_amd/libseunet_hip_base.so; else unset SEUNET_LIB; fi
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-secondary --config none > gpurun_out/r04/ab3_$tag.json 2> gpurun_out/r04/ab3_$tag.err
  python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/r04/ab3_$tag.json') if l.startswith('{')][-1]); print('$tag', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), d['class_ms_per_step'])"
done
