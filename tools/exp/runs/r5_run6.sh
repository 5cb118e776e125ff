export TMPDIR=/tmp
O=gpurun_out/r5j
mkdir -p $O
timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_modules_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
tail -15 $O/tests.txt
for rep in 1 2; do
  for sp in 0 1; do
    PN2_SPLIT=$sp timeout 300 python tools/bench_kernels.py fwd --only 262144,131072 > $O/fwd_split${sp}_$rep.txt 2>&1
    PN2_SPLIT=$sp timeout 300 python tools/bench_kernels.py dgrad --only 262144,131072 > $O/dgrad_split${sp}_$rep.txt 2>&1
  done
done
grep -H "196, 256\|128, 196\|128, 256\|131072, 128, 128" $O/fwd_split*.txt | sed 's/.*fwd_//'
grep -H "196, 128\|131072, 128, 128" $O/dgrad_split*.txt | sed 's/.*dgrad_//'
bash tools/exp/ab_step.sh $O/ab_split.txt "PN2_SPLIT=0" "PN2_SPLIT=1"
