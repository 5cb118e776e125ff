#!/bin/bash
mkdir -p gpurun_out/r6i
echo "== FPS tests with the pieced form"; PN2_FPS_PIECE=128 python3 -m pytest tests/test_geometry_gpu.py -q -k "fps" 2>&1 | tail -3
echo "== same-device two ranks"; python3 -m pytest tests/test_parallel_gpu.py -q -k "two_ranks_on_one_device" 2>&1 | tail -3; cat gpurun_out/rccl_same_device.txt | head -30
for rep in 1 2; do for pc in 0 64 128 256; do for w in msg ssg; do
  PN2_FPS_PIECE=$pc python3 bench.py --workload $w --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('FPS_PIECE=$pc', '$w', d['ms_per_step'])"
done; done; done
export TMPDIR=/tmp
for pc in 0 128; do
  ( cd /tmp && PN2_FPS_PIECE=$pc rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r6i/trace_$pc -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-other-configs --steps 20 --warmup 5 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r6i/trace_$pc.err )
  python3 tools/step_timeline.py $(find gpurun_out/r6i/trace_$pc -name "*kernel_trace.csv" | head -1) --dump 1 > gpurun_out/r6i/step_timeline_piece$pc.txt 2>&1
  rm -rf gpurun_out/r6i/trace_$pc
  head -30 gpurun_out/r6i/step_timeline_piece$pc.txt
done
