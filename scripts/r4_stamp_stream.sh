set -e
cd "$GRAFT_REPO_ROOT"
export SEUNET_LIB=$PWD/se-unet-airseg_amd/libseunet_hip_stamp.so
for L in ec3 dc6 ec2; do
  SEUNET_STAMP=stream REPS=5 WHICH=fwd,dgrad timeout -k 10 120 python3 scripts/bench_conv.py $L 2>&1 | grep -v amdgpu.ids | grep -E "stamps|fwd"
done
