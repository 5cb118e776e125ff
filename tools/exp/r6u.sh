#!/bin/bash
for rep in 1 2 3 4; do
  for f in top sa1 sa2 loss; do
    PN2_BENCH_FORK=$f python3 bench.py --workload msg --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], sys.argv[2], d['ms_per_step'])" $f msg
  done
done
