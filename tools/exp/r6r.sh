#!/bin/bash
for i in 1 2; do python3 tools/exp/fps_concurrency_stress2.py 2>&1 | grep "pooled"; done
python3 tools/exp/fps_concurrency_stress3.py 2>&1 | tail -1
python3 tools/bench_fps.py 2>&1 | head -5
python3 -m pytest tests/test_geometry_gpu.py tests/test_modules_gpu.py -m gpu -q -x 2>&1 | tail -2
for w in msg ssg sa; do python3 bench.py --workload $w --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d['ms_per_step'])" $w; done
