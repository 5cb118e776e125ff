export TMPDIR=/tmp
O=gpurun_out/r5l
mkdir -p $O
timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_modules_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
tail -4 $O/tests.txt
for rep in 1 2; do
  for sp in 0 1; do
    PN2_SPLIT=$sp timeout 300 python tools/bench_kernels.py wgrad --only 262144,131072 > $O/wgrad_split${sp}_$rep.txt 2>&1
  done
  PN2_SPLIT=1 PN2_SPLIT_NARROW=0 timeout 300 python tools/bench_kernels.py fwd --only 1048576,524288 > $O/fwd_narrow0_$rep.txt 2>&1
  PN2_SPLIT=1 PN2_SPLIT_NARROW=1 timeout 300 python tools/bench_kernels.py fwd --only 1048576,524288 > $O/fwd_narrow1_$rep.txt 2>&1
done
grep -H "256, 196\|256, 128\|196, 128" $O/wgrad_split*.txt | sed 's/.*wgrad_//'
grep -H "fwd" $O/fwd_narrow*.txt | sed 's/.*fwd_//'
PN2_SPLIT_NARROW=1 timeout 900 python -m pytest tests/test_mlp_gpu.py -x -q -m gpu 2>&1 | tail -2
bash tools/exp/ab_step.sh $O/ab_split.txt "PN2_SPLIT=0" "PN2_SPLIT=1" "PN2_SPLIT=1 PN2_SPLIT_WGRAD=0"
