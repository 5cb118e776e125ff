export TMPDIR=/tmp
O=gpurun_out/r5aa
mkdir -p $O
timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_modules_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
tail -4 $O/tests.txt
for e in 1 0; do
  echo "== PN2_SPLIT_RES=$e"
  PN2_SPLIT_RES=$e timeout 600 python tools/bench_kernels.py bwd 2>/dev/null | tee $O/bwd_$e.txt | tail -30
done
bash tools/exp/ab_step.sh $O/ab.txt "PN2_SPLIT_RES=0" "-" "PN2_SPLIT_RES=0" "-" > /dev/null
sort $O/ab.txt
