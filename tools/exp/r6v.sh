#!/bin/bash
# the headline's files again after the fork default moved behind sa2 (bench.py only: the kernel sources and their PMC files stand)
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r06d; mkdir -p $O
python3 bench.py > $O/bench_msg.json 2> $O/bench_msg.err
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_msg -o t -- python3 $R/bench.py --no-cpu-baseline > /dev/null 2> $O/prof_msg.err )
f=$(find $O/prof_msg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/prof_msg_kernel_stats.csv; rm -rf $O/prof_msg
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/trace_msg -o t -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 > /dev/null 2> $O/trace_msg.err )
python3 tools/step_timeline.py $(find $O/trace_msg -name "*kernel_trace.csv" | head -1) --dump 1 > $O/step_timeline_msg.txt 2>&1
rm -rf $O/trace_msg
python3 -m pytest tests/test_modules_gpu.py -m gpu -q -x 2>&1 | tail -1
python3 -c "import json; d=json.loads(open('$O/bench_msg.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'])"
