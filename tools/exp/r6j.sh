#!/bin/bash
# ABBA of SPLIT_WG2 (four-wave split forward kernels as two workgroups per CU), MSG cfg3 and cfg5
run() { env "$@" python3 bench.py --no-cpu-baseline --no-other-configs --no-roofline $EXTRA 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', '$EXTRA', d['ms_per_step'])"; }
EXTRA=""
for rep in 1 2 3; do run PN2_SPLIT_WG2=0; run PN2_SPLIT_WG2=1; run PN2_SPLIT_WG2=1; run PN2_SPLIT_WG2=0; done
for rep in 1 2; do run PN2_SPLIT_WG2=0 PN2_POOL_CF=2; run PN2_SPLIT_WG2=1 PN2_POOL_CF=2; run PN2_SPLIT_WG2=1 PN2_POOL_CF=0; run PN2_SPLIT_WG2=0 PN2_POOL_CF=0; done
EXTRA="--workload msg --points 65536 --batch 8 --npoint-scale 16 --steps 5 --warmup 2"
for rep in 1 2; do run PN2_SPLIT_WG2=0; run PN2_SPLIT_WG2=1; run PN2_SPLIT_WG2=1 PN2_POOL_CF=2; run PN2_SPLIT_WG2=0 PN2_POOL_CF=0; done
