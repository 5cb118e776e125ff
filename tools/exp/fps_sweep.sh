python -m pytest tests/test_geometry_gpu.py -x -q -k "fps or farthest" 2>&1 | tail -2
for m in 0 1 2; do echo "== ROWMAP $m"; PN2_FPS_ROWMAP=$m python tools/bench_fps.py 2>&1 | grep "N= *\(16384\|20000\|25000\|28672\)"; done
for m in 1 2; do echo "== ROWMAP $m rows from PPT 16"; PN2_FPS_ROWS_MIN_PPT=16 PN2_FPS_ROWMAP=$m python tools/bench_fps.py 2>&1 | grep "N= *\(16384\|20000\)"; done
