export TMPDIR=/tmp
O=gpurun_out/r5h
mkdir -p $O
timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_modules_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
tail -4 $O/tests.txt
PN2_RING=1 timeout 900 python -m pytest tests/test_mlp_gpu.py -x -q -m gpu > $O/tests_ring.txt 2>&1
tail -2 $O/tests_ring.txt
timeout 600 python bench.py > $O/bench_msg.json 2> $O/bench_msg.err
python3 -c "
import json; d=json.load(open('$O/bench_msg.json')); print(d['ms_per_step'], d['roofline']['frac'], json.dumps(d.get('other_configs'), indent=0))"
