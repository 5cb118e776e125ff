export TMPDIR=/tmp
O=gpurun_out/r5av
mkdir -p $O
PN2_WGRAD_TWO_PHASE=1 PN2_WGRAD_TWO_PHASE_ALL=1 timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_modules_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
tail -2 $O/tests.txt
for e in "PN2_WGRAD_TWO_PHASE=0" "PN2_WGRAD_TWO_PHASE=1 PN2_WGRAD_TWO_PHASE_ALL=1" "PN2_WGRAD_TWO_PHASE=0" "PN2_WGRAD_TWO_PHASE=1 PN2_WGRAD_TWO_PHASE_ALL=1"; do
  echo "== $e"
  env $e timeout 300 python tools/bench_kernels.py wgrad --only 262144,131072 2>/dev/null | grep "256, 196\|256, 128\|196, 128"
done
bash tools/exp/ab_step.sh $O/ab.txt "-" "PN2_WGRAD_TWO_PHASE=1" "PN2_WGRAD_TWO_PHASE=1 PN2_WGRAD_TWO_PHASE_ALL=1" "-" "PN2_WGRAD_TWO_PHASE=1" "PN2_WGRAD_TWO_PHASE=1 PN2_WGRAD_TWO_PHASE_ALL=1" > /dev/null
sort $O/ab.txt
