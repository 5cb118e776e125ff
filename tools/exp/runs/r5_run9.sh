export TMPDIR=/tmp
O=gpurun_out/r5m
mkdir -p $O
timeout 900 python -m pytest tests/test_mlp_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
tail -3 $O/tests.txt
PN2_SPLIT_WGRAD=2 timeout 900 python -m pytest tests/test_mlp_gpu.py -x -q -m gpu > $O/tests_bp32.txt 2>&1
tail -3 $O/tests_bp32.txt
PN2_SPLIT_NARROW=1 timeout 900 python -m pytest tests/test_mlp_gpu.py -x -q -m gpu > $O/tests_narrow.txt 2>&1
grep -A40 "^___" $O/tests_narrow.txt | head -80
tail -3 $O/tests_narrow.txt
for rep in 1 2; do
  for v in "PN2_SPLIT=0" "PN2_SPLIT_WGRAD=1" "PN2_SPLIT_WGRAD=2"; do
    env $v timeout 300 python tools/bench_kernels.py wgrad --only 262144,131072 2>/dev/null | grep "256, 196\|256, 128\|196, 128" | sed "s/^/$v /"
  done
done
