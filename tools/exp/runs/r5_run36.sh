export TMPDIR=/tmp
O=gpurun_out/r5at
mkdir -p $O
PN2_LIB_PATH=pointnet12_amd/libpn2_hip_wall.so timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_modules_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
tail -2 $O/tests.txt
for lib in libpn2_hip.so libpn2_hip_wall.so; do
  echo "== $lib"
  PN2_LIB_PATH=pointnet12_amd/$lib timeout 300 python tools/bench_kernels.py dgrad --only 262144,131072 2>/dev/null | grep "256, 196\|256, 128"
  PN2_LIB_PATH=pointnet12_amd/$lib timeout 300 python tools/bench_kernels.py fwd --only 1048576 2>/dev/null | grep "96, 128\|64, 128"
done
bash tools/exp/ab_step.sh $O/ab.txt "-" "PN2_LIB_PATH=pointnet12_amd/libpn2_hip_wall.so" "PN2_LIB_PATH=pointnet12_amd/libpn2_hip_wks2.so" "-" "PN2_LIB_PATH=pointnet12_amd/libpn2_hip_wall.so" "PN2_LIB_PATH=pointnet12_amd/libpn2_hip_wks2.so" > /dev/null
sort $O/ab.txt
