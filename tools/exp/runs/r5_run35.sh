export TMPDIR=/tmp
O=gpurun_out/r5ao
mkdir -p $O
timeout 900 python -m pytest tests/test_mlp_gpu.py -x -q -m gpu -k "bf16_split" > $O/tests.txt 2>&1
tail -25 $O/tests.txt
