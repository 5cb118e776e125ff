#!/bin/bash
python3 -m pytest tests/test_mlp_gpu.py -q -x -k "fixed_operands or drift_with or first_layers_sums or fused_first_layer_option or (shared_mlp and (262144 or 131072))" 2>&1 | tail -3
echo "== DXFREE (product)"; python3 tools/bench_kernels.py bwd --only 1048576,524288 2>&1 | grep "^bwd"
echo "== all waves stage (variant)"; PN2_LIB_PATH=$PWD/pointnet12_amd/libpn2_hip_nodxfree.so python3 tools/bench_kernels.py bwd --only 1048576,524288 2>&1 | grep "^bwd"
run() { env "$@" python3 bench.py --no-cpu-baseline --no-other-configs --no-roofline $EXTRA 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', '$EXTRA', d['ms_per_step'])"; }
V=PN2_LIB_PATH=$PWD/pointnet12_amd/libpn2_hip_nodxfree.so
EXTRA=""
for rep in 1 2 3; do run X=1; run $V; run $V; run X=1; done
for rep in 1 2; do run PN2_FUSE_FIRST=1; run PN2_FUSE_FIRST=0; done
EXTRA="--workload ssg"
for rep in 1 2; do run X=1; run $V; done
EXTRA="--workload msg --points 65536 --batch 8 --npoint-scale 16 --steps 5 --warmup 2"
for rep in 1 2; do run X=1; run $V; done
