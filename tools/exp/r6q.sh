#!/bin/bash
# same-box A/B of the PN2_LDS_SETTLE guard: the product library against a build with the macro empty, alternating
for rep in 1 2 3; do
  for lib in product nosettle; do
    for w in msg ssg sa; do
      if [ $lib = product ]; then unset PN2_LIB_PATH; else export PN2_LIB_PATH=$PWD/pointnet12_amd/libpn2_hip_nosettle.so; fi
      python3 bench.py --workload $w --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], sys.argv[2], d['ms_per_step'])" $lib $w
    done
  done
done
