#!/bin/bash
for rep in 1 2 3 4 5 6 7 8; do
  for v in 0 1; do
    PN2_FUSE_FIRST=$v python3 bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('msg', sys.argv[1], d['ms_per_step'])" $v
  done
done
for rep in 1 2; do
  for v in 0 1; do
    PN2_FUSE_FIRST=$v python3 bench.py --workload msg --points 65536 --batch 8 --npoint-scale 16 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg5msg', sys.argv[1], d['ms_per_step'])" $v
    PN2_FUSE_FIRST=$v python3 bench.py --workload ssg --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ssg', sys.argv[1], d['ms_per_step'])" $v
  done
done
