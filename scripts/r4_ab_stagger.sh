set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
timeout -k 10 600 python3 -m pytest tests/test_ops_gpu.py -q -x -k "stream" > gpurun_out/r04/stag_ops.log 2>&1 || { tail -30 gpurun_out/r04/stag_ops.log; exit 1; }
tail -1 gpurun_out/r04/stag_ops.log
for L in ec3 dc6 ec2 ec1; do
  for tag in base new base new; do
    if [ $tag = base ]; then export SEUNET_LIB=$PWD/se-unet-airseg_amd/libseunet_hip_base.so; else unset SEUNET_LIB; fi
    echo -n "$L $tag  "; REPS=10 WHICH=fwd,dgrad timeout -k 10 120 python3 scripts/bench_conv.py $L 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/.*| STREAM fwd/STREAM fwd/'
  done
done
export SEUNET_LIB=$PWD/se-unet-airseg_amd/libseunet_hip_stamp.so
for L in ec3 dc6; do
  SEUNET_STAMP=stream REPS=5 WHICH=fwd,dgrad timeout -k 10 120 python3 scripts/bench_conv.py $L 2>&1 | grep -E "stamps"
done
