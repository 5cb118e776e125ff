export TMPDIR=/tmp
O=gpurun_out/r5n
mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/tests_gpu.txt 2>&1
tail -4 $O/tests_gpu.txt
timeout 600 python bench.py > $O/bench_msg.json 2> $O/bench_msg.err
python3 -c "
import json; d=json.load(open('$O/bench_msg.json')); print(d['ms_per_step'], d['dtype'][:40], d['roofline']['kernel'], d['roofline']['frac'])
for f in d['roofline']['families']: print(' ', f['name'], f['ms_per_step'], f['frac'])
for k,v in d['other_configs'].items(): print(k, v.get('ms_per_step'))"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
