#!/bin/bash
for rep in 1 2 3 4 5 6; do
  for v in 0 1; do
    PN2_FUSE_FIRST=$v python3 bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('msg', sys.argv[1], d['ms_per_step'])" $v
  done
done
python3 tools/exp/split_bias_probe.py 2>/dev/null | head -3 >/dev/null
grep -m1 "clk" /dev/null
