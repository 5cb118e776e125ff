export TMPDIR=/tmp
O=gpurun_out/r5k
mkdir -p $O
timeout 1500 python -m pytest tests/test_parity_stages_gpu.py tests/test_parity_fullsize_gpu.py -x -q -m gpu > $O/tests_parity.txt 2>&1
tail -5 $O/tests_parity.txt
cp gpurun_out/parity_stages.json $O/parity_stages.json 2>/dev/null
cp gpurun_out/parity_fullsize.json $O/parity_fullsize.json 2>/dev/null
