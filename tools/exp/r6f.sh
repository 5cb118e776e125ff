#!/bin/bash
mkdir -p gpurun_out/r6f
python3 -m pytest tests/test_mlp_gpu.py -q -x -k "pooled_last_layer_backward or negative_and_zero" 2>&1 | tail -4
tools/exp/prof_cmd.sh r6f tools/bench_kernels.py bwdcf 2>&1 | grep "split_bwd_cf\|cf_finish\|cf_prep"
echo "== nostore probe, SPLIT_WG2=0"; python3 tools/exp/nostore_probe.py 2>&1 | grep fwdpool
echo "== nostore probe, SPLIT_WG2=1"; PN2_SPLIT_WG2=1 python3 tools/exp/nostore_probe.py 2>&1 | grep fwdpool
echo "== fwd microbench rows of sa1, WG2=0 / 1"
for w in 0 1; do PN2_SPLIT_WG2=$w python3 tools/bench_kernels.py fwd --only 1048576,524288 2>&1 | grep "^fwd "; done
for w in 0 1; do echo "== bench msg SPLIT_WG2=$w"; PN2_SPLIT_WG2=$w python3 bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['bound'], d['roofline']['frac'], d['roofline']['avg_us'])"; done
PN2_POOL_CF=0 python3 bench.py --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('POOL_CF=0', d['ms_per_step'])"
PN2_POOL_CF=2 python3 bench.py --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('POOL_CF=2', d['ms_per_step'])"
python3 bench.py --no-cpu-baseline --no-other-configs > gpurun_out/r6f/bench.json 2> gpurun_out/r6f/bench.err; tail -3 gpurun_out/r6f/bench.err
