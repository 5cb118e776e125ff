export TMPDIR=/tmp
O=gpurun_out/r5ae
mkdir -p $O
timeout 900 python -m pytest tests/test_mlp_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
tail -2 $O/tests.txt
for e in 2 0; do
  echo "== PN2_SPLIT_RES=$e"
  PN2_SPLIT_RES=$e timeout 600 python tools/bench_kernels.py bwd 2>/dev/null | grep -v "65536\|131072"
done
PN2_SPLIT_RES=2 PN2_LIB_PATH=pointnet12_amd/libpn2_hip_stamp.so timeout 300 python tools/stamp_res.py 2>&1 | grep -v "^fwd\|transform" > $O/stamp.txt
grep -A9 "^bwd (1048576, 128, 96\|^bwd (524288, 32, 32" $O/stamp.txt | cut -c1-200
bash tools/exp/ab_step.sh $O/ab.txt "PN2_SPLIT_RES=0" "-" "PN2_SPLIT_RES=2" "PN2_SPLIT_RES=0" "-" "PN2_SPLIT_RES=2" > /dev/null
sort $O/ab.txt
