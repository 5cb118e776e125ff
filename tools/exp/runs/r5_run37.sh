export TMPDIR=/tmp
O=gpurun_out/r5au
mkdir -p $O
timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_modules_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
tail -2 $O/tests.txt
for lib in libpn2_hip.so libpn2_hip_tnold.so libpn2_hip.so libpn2_hip_tnold.so; do
  echo "== $lib"
  PN2_LIB_PATH=pointnet12_amd/$lib timeout 300 python tools/bench_kernels.py wgrad --only 262144,131072 2>/dev/null | grep "256, 196\|256, 128\|196, 128"
done
bash tools/exp/ab_step.sh $O/ab.txt "-" "PN2_LIB_PATH=pointnet12_amd/libpn2_hip_tnold.so" "-" "PN2_LIB_PATH=pointnet12_amd/libpn2_hip_tnold.so" > /dev/null
sort $O/ab.txt
