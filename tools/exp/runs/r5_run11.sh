export TMPDIR=/tmp
O=gpurun_out/r5o
mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/tests_gpu.txt 2>&1
tail -4 $O/tests_gpu.txt
for rep in 1 2; do
  for v in "PN2_SPLIT=0" "PN2_SPLIT_K256=0" "PN2_SPLIT_K256=1"; do
    env $v timeout 300 python tools/bench_kernels.py dgrad --only 262144,131072 2>/dev/null | grep "256, 196\|256, 128" | sed "s/^/$v /"
  done
done
timeout 600 python bench.py > $O/bench_msg.json 2> $O/bench_msg.err
python3 -c "
import json; d=json.load(open('$O/bench_msg.json')); print(d['ms_per_step'], d['dtype'][:40], d['roofline']['kernel'], d['roofline']['frac'])
for f in d['roofline']['families']: print(' ', f['name'], f['ms_per_step'], f['frac'])
for k,v in d['other_configs'].items(): print(k, v.get('ms_per_step'))"
bash tools/exp/ab_step.sh $O/ab.txt "PN2_SPLIT=0" "PN2_SPLIT_K256=0" "-"
