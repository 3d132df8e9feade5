set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
for i in 1 2; do
timeout -k 10 600 python3 -m pytest tests/test_ops_gpu.py -q -x -k "stream" > gpurun_out/r04/sg_ops.log 2>&1 || { tail -30 gpurun_out/r04/sg_ops.log; exit 1; }
tail -1 gpurun_out/r04/sg_ops.log
done
for L in ec3 dc6 ec2; do
  for tag in direct staged direct staged; do
    if [ $tag = direct ]; then export SEUNET_STREAM_NO_STAGE=1; else unset SEUNET_STREAM_NO_STAGE; fi
    echo -n "$L $tag  "; REPS=10 WHICH=fwd,dgrad timeout -k 10 120 python3 scripts/bench_conv.py $L 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/.*| STREAM fwd/STREAM fwd/'
  done
done
unset SEUNET_STREAM_NO_STAGE
timeout -k 10 900 python3 -m pytest tests/test_net_gpu.py -q -x -k "16bit_modes_against or full_size_properties or bitwise_reproducible or prior_contents or config4 or same_choice or sliding_window_192 or shares_the_gpu" > gpurun_out/r04/sg_net.log 2>&1 || { tail -30 gpurun_out/r04/sg_net.log; exit 1; }
tail -1 gpurun_out/r04/sg_net.log
for tag in direct staged direct staged; do
  if [ $tag = direct ]; then export SEUNET_STREAM_NO_STAGE=1; else unset SEUNET_STREAM_NO_STAGE; fi
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline > gpurun_out/r04/ab8_$tag.json 2> gpurun_out/r04/ab8_$tag.err
  python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/r04/ab8_$tag.json') if l.startswith('{')][-1]); c=d['class_ms_per_step']; print('$tag', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), 'conv_fwd', c['conv_fwd'], 'dgrad', c['dgrad'], 'wgrad', c['wgrad'], 'window512', d['window512']['seconds'])"
done
