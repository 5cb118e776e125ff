#!/bin/bash
for rep in 1 2 3; do
  for v in 0 8192 16384 65536; do
    for w in msg ssg; do
      PN2_WGRAD_SIDE_MAX_ROWS=$v python3 bench.py --workload $w --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[2], sys.argv[1], d['ms_per_step'])" $v $w
    done
  done
done
