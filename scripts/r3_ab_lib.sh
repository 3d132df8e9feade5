# same-box A/B of two builds of the library on single conv launches: base = se-unet-airseg_amd/libseunet_hip_base.so
mkdir -p gpurun_out
for L in dc5 dc3 dc4 ec4 ec5 ec6; do
  for tag in base new; do
    if [ $tag = base ]; then export SEUNET_LIB=$PWD/se-unet-airseg_amd/libseunet_hip_base.so; else unset SEUNET_LIB; fi
    echo -n "$tag  "; REPS=10 WHICH=${WHICH:-fwd,dgrad} timeout -k 10 120 python scripts/bench_conv.py $L 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
