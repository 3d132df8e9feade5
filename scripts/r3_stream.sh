mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -x -k "stream" > gpurun_out/r3_stream_ops.log 2>&1
echo "ops rc=$?"; tail -3 gpurun_out/r3_stream_ops.log
for L in ec3 dc6 ec2 ec1; do
  for tag in base new; do
    if [ $tag = base ]; then export SEUNET_LIB=$PWD/se-unet-airseg_amd/libseunet_hip_base.so; else unset SEUNET_LIB; fi
    echo -n "$tag  "; REPS=10 WHICH=fwd,dgrad timeout -k 10 120 python scripts/bench_conv.py $L 2>&1 | grep -v amdgpu.ids | tail -1 | grep -o "^[a-z0-9]* \|STREAM fwd [0-9.]* ms\|STREAM dgrad+= [0-9.]* ms" | paste - - -
  done
done
