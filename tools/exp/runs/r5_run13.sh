export TMPDIR=/tmp
O=gpurun_out/r5q
mkdir -p $O
timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_modules_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
tail -3 $O/tests.txt
PN2_LIB_PATH=pointnet12_amd/libpn2_hip_stamp.so timeout 300 python tools/stamp_wide.py > $O/stamp_split.txt 2>&1
grep -v "wave [1235679]" $O/stamp_split.txt
for rep in 1 2; do
  timeout 300 python tools/bench_kernels.py fwd --only 262144,131072 2>/dev/null | grep "196, 256)\|128, 196)\|128, 256)\|131072, 128, 128"
  timeout 300 python tools/bench_kernels.py dgrad --only 262144,131072 2>/dev/null | grep "256, 196\|256, 128\|196, 128\|131072, 128, 128"
done
bash tools/exp/ab_step.sh $O/ab.txt "PN2_SPLIT=0" "-"
