# same-box A/B of two library builds on isolated marching-conv launches: base = se-unet-airseg_amd/libseunet_hip_base.so
# (a copy of the previous build; not kept in the tree)
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
timeout -k 10 600 python3 -m pytest tests/test_ops_gpu.py -q -x -k "march" > gpurun_out/r04/march_ops.log 2>&1 || { tail -30 gpurun_out/r04/march_ops.log; exit 1; }
tail -1 gpurun_out/r04/march_ops.log
for L in dc5 dc3 dc4 ec6 ec4 ec8; do
  for tag in base new base new; do
    if [ $tag = base ]; then export SEUNET_LIB=$PWD/se-unet-airseg_amd/libseunet_hip_base.so; else unset SEUNET_LIB; fi
    echo -n "$L $tag  "; REPS=10 WHICH=fwd,dgrad timeout -k 10 120 python3 scripts/bench_conv.py $L 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/.*| MARCH fwd/MARCH fwd/; s/^.*B4: fwd [^|]*| dgrad [^|]*| //'
  done
done
