export TMPDIR=/tmp
R=$(pwd)
O=$R/gpurun_out/r5v
mkdir -p $O
timeout 900 python -m pytest tests/test_modules_gpu.py tests/test_parallel_gpu.py tests/test_train_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
tail -2 $O/tests.txt
for w in msg ssg; do python3 bench.py --workload $w --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', d['ms_per_step'])"; done
for w in msg ssg; do python3 bench.py --workload $w --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', d['ms_per_step'])"; done
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/prof_msg -o t -- python3 $R/bench.py --workload msg --no-cpu-baseline --no-roofline --steps 20 --warmup 5 > $O/bench_prof.json 2> $O/prof.err )
python3 tools/step_timeline.py $O/prof_msg/t_kernel_trace.csv --dump 1 > $O/timeline.txt 2>&1
head -24 $O/timeline.txt
