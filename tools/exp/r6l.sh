#!/bin/bash
python3 -m pytest tests/test_mlp_gpu.py -q -s -x -k "first_layers_sums or shared_mlp or first_layer_weight" 2>&1 | grep "bwd_first\|passed\|failed\|Error" | tail -8
python3 -m pytest tests/test_parity_stages_gpu.py tests/test_modules_gpu.py -q -x 2>&1 | tail -2
run() { env "$@" python3 bench.py --no-cpu-baseline --no-other-configs --no-roofline $EXTRA 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', '$EXTRA', d['ms_per_step'])"; }
EXTRA=""
for rep in 1 2 3; do run PN2_FUSE_FIRST=0; run PN2_FUSE_FIRST=1; run PN2_FUSE_FIRST=1; run PN2_FUSE_FIRST=0; done
EXTRA="--workload msg --points 65536 --batch 8 --npoint-scale 16 --steps 5 --warmup 2"
for rep in 1 2; do run PN2_FUSE_FIRST=0; run PN2_FUSE_FIRST=1; done
export PN2_MSG_STREAMS=0
for f in 0 1; do echo "== FUSE_FIRST=$f"; PN2_FUSE_FIRST=$f tools/exp/prof_cmd.sh r6l_f$f bench.py --no-graph --no-roofline --no-cpu-baseline --no-other-configs 2>&1 | grep "split_bwd_res_kernel<[23], 2, false, true\|wgrad_first_cf"; done
