#!/bin/bash
# CUs reserved for the geometry branch while sa1's forward runs (PN2_BENCH_RESERVE -> library option RESERVE_CUS), ABBA
run() { env "$@" python3 bench.py --no-cpu-baseline --no-other-configs --no-roofline $EXTRA 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', '$EXTRA', d['ms_per_step'])"; }
EXTRA=""
for rep in 1 2; do for r in 0 16 24 32 16 0; do run PN2_BENCH_RESERVE=$r; done; done
EXTRA="--workload ssg"
for r in 0 16 32 0; do run PN2_BENCH_RESERVE=$r PN2_BENCH_FORK=top; done
python3 -m pytest tests/test_modules_gpu.py -q -k "prefetched_geometry_graph" 2>&1 | tail -3
python3 tools/bench_kernels.py pair 2>&1 | tail -8
