export TMPDIR=/tmp
O=gpurun_out/r5ag
mkdir -p $O
timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_modules_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
tail -2 $O/tests.txt
for lib in libpn2_hip.so libpn2_hip_noacc2.so; do
  echo "== $lib"
  for rep in 1 2; do
  PN2_LIB_PATH=pointnet12_amd/$lib timeout 300 python tools/bench_kernels.py fwd --only 131072 2>/dev/null | grep "131072, 128, 128"
  PN2_LIB_PATH=pointnet12_amd/$lib timeout 300 python tools/bench_kernels.py dgrad --only 262144,131072 2>/dev/null | grep "196, 128\|131072, 128, 128"
  done
done
bash tools/exp/ab_step.sh $O/ab.txt "PN2_LIB_PATH=pointnet12_amd/libpn2_hip_noacc2.so" "-" "PN2_LIB_PATH=pointnet12_amd/libpn2_hip_noacc2.so" "-" > /dev/null
sort $O/ab.txt
