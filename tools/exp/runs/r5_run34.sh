export TMPDIR=/tmp
O=gpurun_out/r5an
mkdir -p $O
PN2_SPLIT_MIN_ROWS_128=65536 timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_modules_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
tail -2 $O/tests.txt
bash tools/exp/ab_step.sh $O/ab.txt "PN2_SPLIT_MIN_ROWS_128=65536" "-" "PN2_SPLIT_MIN_ROWS_128=65536" "-" "PN2_SPLIT_MIN_ROWS_128=65536" "-" > /dev/null
sort $O/ab.txt
