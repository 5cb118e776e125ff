#!/bin/bash
# soak: pn2_fps beside the MLP kernels on the final library, and the captured-vs-eager geometry comparison repeated
{
echo "# final library (PN2_OPAQUE): pn2_fps on a side stream beside ..., results against the same launch alone"
python3 - <<'PY'
import sys
sys.argv = ["x"]
sys.path.insert(0, "tools/exp")
import fps_concurrency_stress2 as s2
s2.main(trials=600)
PY
python3 tools/exp/fps_concurrency_stress3.py 2>&1 | tail -1
echo "# test_prefetched_geometry_graph_matches_eager, 30 runs in fresh processes"
ok=0; for i in $(seq 1 30); do python3 -m pytest tests/test_modules_gpu.py -m gpu -q -x -k prefetched_geometry_graph_matches_eager -p no:cacheprovider 2>&1 | tail -1 | grep -q "1 passed" && ok=$((ok+1)); done; echo "passed $ok of 30"
} 2>&1 | grep -v amdgpu.ids > gpurun_out/fps_coresidency_soak.txt
cat gpurun_out/fps_coresidency_soak.txt
