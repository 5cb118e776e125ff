export TMPDIR=/tmp
O=gpurun_out/r5ac
mkdir -p $O
timeout 900 python -m pytest tests/test_mlp_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
tail -2 $O/tests.txt
for e in 1 0 1 0; do
  echo "== PN2_SPLIT_RES=$e"
  PN2_SPLIT_RES=$e timeout 600 python tools/bench_kernels.py bwd 2>/dev/null | grep -v "65536\|131072"
done
