export TMPDIR=/tmp
O=gpurun_out/r5ah
mkdir -p $O
PN2_SPLIT_K256=2 timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_modules_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
tail -2 $O/tests.txt
for e in 2 1 2 1; do
  echo "== PN2_SPLIT_K256=$e"
  PN2_SPLIT_K256=$e timeout 300 python tools/bench_kernels.py dgrad --only 262144 2>/dev/null | grep "256, 196"
done
bash tools/exp/ab_step.sh $O/ab.txt "PN2_SPLIT_K256=2" "-" "PN2_SPLIT_K256=2" "-" > /dev/null
sort $O/ab.txt
