set -e
cd "$GRAFT_REPO_ROOT"
for L in ec3 dc6 ec2; do
  for tag in full nostats nostores nomfma nofrag; do
    case $tag in full) unset SEUNET_LIB;; nostats) export SEUNET_LIB=$PWD/se-unet-airseg_amd/libseunet_hip_probe1.so;; nostores) export SEUNET_LIB=$PWD/se-unet-airseg_amd/libseunet_hip_probe2.so;; nomfma) export SEUNET_LIB=$PWD/se-unet-airseg_amd/libseunet_hip_probe3.so;; nofrag) export SEUNET_LIB=$PWD/se-unet-airseg_amd/libseunet_hip_probe4.so;; esac
    echo -n "$L $tag  "; REPS=10 WHICH=fwd,dgrad timeout -k 10 120 python3 scripts/bench_conv.py $L 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/.*| STREAM fwd/STREAM fwd/'
  done
done
