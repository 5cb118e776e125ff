export TMPDIR=/tmp
O=gpurun_out/r5bd
mkdir -p $O
timeout 900 python -m pytest tests/test_geometry_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
tail -2 $O/tests.txt
timeout 300 python tools/exp/fps_xcd_debug.py 2>&1 | tail -14
for e in 1 0 1 0; do
  PN2_FPS_XCD_DEAL=$e python3 bench.py --workload msg --points 65536 --batch 8 --npoint-scale 16 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 msg deal=$e', d['ms_per_step'])"
done
