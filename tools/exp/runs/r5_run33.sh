export TMPDIR=/tmp
O=gpurun_out/r5al
mkdir -p $O
timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_modules_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
tail -2 $O/tests.txt
for lib in libpn2_hip.so libpn2_hip_nopin.so libpn2_hip.so libpn2_hip_nopin.so; do
  echo "== $lib"
  PN2_LIB_PATH=pointnet12_amd/$lib timeout 300 python tools/bench_kernels.py dgrad --only 262144,131072 2>/dev/null | grep "256, 196\|256, 128\|196, 128\|131072, 128, 128"
done
PN2_LIB_PATH=pointnet12_amd/libpn2_hip_stamp.so timeout 300 python tools/stamp_wide.py > $O/stamp.txt 2>&1
grep -A9 "^dgrad (262144, 256" $O/stamp.txt | cut -c1-220
bash tools/exp/ab_step.sh $O/ab.txt "PN2_LIB_PATH=pointnet12_amd/libpn2_hip_nopin.so" "-" "PN2_LIB_PATH=pointnet12_amd/libpn2_hip_nopin.so" "-" > /dev/null
sort $O/ab.txt
for lib in libpn2_hip.so libpn2_hip_nopin.so; do
  echo "== fwd $lib"
  PN2_LIB_PATH=pointnet12_amd/$lib timeout 300 python tools/bench_kernels.py fwd --only 1048576,262144,131072 2>/dev/null | grep "196, 256)\|128, 196)\|128, 256)\|131072, 128, 128\|1048576, 96, 128\|1048576, 64, 96\|64, 128"
done
