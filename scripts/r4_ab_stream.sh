# same-box A/B of two builds of the library on isolated streaming-conv launches: variant = se-unet-airseg_amd/libseunet_hip_base.so
# (build it from a copy of csrc/ with the change under test, e.g. -DSEUNET_STREAM_ROWS=4; not kept in the tree)
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
SEUNET_LIB=$PWD/se-unet-airseg_amd/libseunet_hip_base.so timeout -k 10 600 python3 -m pytest tests/test_ops_gpu.py -q -x -k "stream" > gpurun_out/r04/stream_ops.log 2>&1 || { tail -30 gpurun_out/r04/stream_ops.log; exit 1; }
tail -2 gpurun_out/r04/stream_ops.log
for L in ec3 dc6 ec2; do
  for tag in variant default variant default; do
    if [ $tag = variant ]; then export SEUNET_LIB=$PWD/se-unet-airseg_amd/libseunet_hip_base.so; else unset SEUNET_LIB; fi
    echo -n "$L $tag  "; REPS=10 WHICH=${WHICH:-fwd,dgrad} timeout -k 10 120 python3 scripts/bench_conv.py $L 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
