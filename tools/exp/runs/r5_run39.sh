export TMPDIR=/tmp
O=gpurun_out/r5aw
mkdir -p $O
PN2_SPLIT_RES_MIN_TILES_128=512 timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_modules_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
tail -2 $O/tests.txt
for e in 4096 512 4096 512; do
  echo "== PN2_SPLIT_RES_MIN_TILES_128=$e"
  PN2_SPLIT_RES_MIN_TILES_128=$e timeout 300 python tools/bench_kernels.py bwd 2>/dev/null | grep "128, 128"
done
bash tools/exp/ab_step.sh $O/ab.txt "-" "PN2_SPLIT_RES_MIN_TILES_128=1024" "PN2_SPLIT_RES_MIN_TILES_128=256" "-" "PN2_SPLIT_RES_MIN_TILES_128=1024" "PN2_SPLIT_RES_MIN_TILES_128=256" > /dev/null
sort $O/ab.txt
