set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
timeout -k 10 600 python3 -m pytest tests/test_ops_gpu.py -q -x -k "stream" > gpurun_out/r04/zs_ops.log 2>&1 || { tail -30 gpurun_out/r04/zs_ops.log; exit 1; }
tail -1 gpurun_out/r04/zs_ops.log
timeout -k 10 900 python3 -m pytest tests/test_net_gpu.py -q -x -k "16bit_modes_against or full_size_properties or bitwise_reproducible or prior_contents or config4 or same_choice or sliding_window_192" > gpurun_out/r04/zs_net.log 2>&1 || { tail -30 gpurun_out/r04/zs_net.log; exit 1; }
tail -1 gpurun_out/r04/zs_net.log
for tag in 1024 256 1024 256; do
  SEUNET_STREAM_WGS=$tag timeout -k 10 300 python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline > gpurun_out/r04/ab7_$tag.json 2> gpurun_out/r04/ab7_$tag.err
  python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/r04/ab7_$tag.json') if l.startswith('{')][-1]); c=d['class_ms_per_step']; print('wgs $tag', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), 'conv_fwd', c['conv_fwd'], 'dgrad', c['dgrad'], 'wgrad', c['wgrad'], 'window512', d['window512']['seconds'])"
done
