#!/bin/bash
export PN2_MSG_STREAMS=0
tot() { python3 - "$1" <<'PY'
import csv,sys
rows=list(csv.DictReader(open("gpurun_out/%s/kernel_stats.csv"%sys.argv[1])))
t=sum(float(r["TotalDurationNs"]) for r in rows if "at::" not in r["Name"] and "rocclr" not in r["Name"])
print("library kernels total per step: %.3f ms"%(t/25/1e6))
PY
}
echo "== A: all waves stage, FUSE_FIRST=0"; PN2_LIB_PATH=$PWD/pointnet12_amd/libpn2_hip_nodxfree.so PN2_FUSE_FIRST=0 tools/exp/prof_cmd.sh r6n_a bench.py --no-graph --no-roofline --no-cpu-baseline --no-other-configs 2>&1 | grep "split_bwd_res_kernel<[23], 2, false, true\|wgrad_first_cf"; tot r6n_a
echo "== B: DXFREE, FUSE_FIRST=1"; PN2_FUSE_FIRST=1 tools/exp/prof_cmd.sh r6n_b bench.py --no-graph --no-roofline --no-cpu-baseline --no-other-configs 2>&1 | grep "split_bwd_res_kernel<[23], 2, false, true\|wgrad_first_cf"; tot r6n_b
echo "== C: DXFREE, FUSE_FIRST=0"; PN2_FUSE_FIRST=0 tools/exp/prof_cmd.sh r6n_c bench.py --no-graph --no-roofline --no-cpu-baseline --no-other-configs 2>&1 | grep "split_bwd_res_kernel<[23], 2, false, true\|wgrad_first_cf"; tot r6n_c
echo "== D: all waves stage, FUSE_FIRST=1"; PN2_LIB_PATH=$PWD/pointnet12_amd/libpn2_hip_nodxfree.so PN2_FUSE_FIRST=1 tools/exp/prof_cmd.sh r6n_d bench.py --no-graph --no-roofline --no-cpu-baseline --no-other-configs 2>&1 | grep "split_bwd_res_kernel<[23], 2, false, true\|wgrad_first_cf"; tot r6n_d
unset PN2_MSG_STREAMS
run() { env "$@" python3 bench.py --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'])"; }
V=PN2_LIB_PATH=$PWD/pointnet12_amd/libpn2_hip_nodxfree.so
for rep in 1 2 3; do run $V PN2_FUSE_FIRST=0; run PN2_FUSE_FIRST=1; run $V PN2_FUSE_FIRST=1; run PN2_FUSE_FIRST=0; done
