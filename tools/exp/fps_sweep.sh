python -m pytest tests/test_geometry_gpu.py -x -q -k "fps or farthest" 2>&1 | tail -2
for f in 0 1; do echo "== FINE $f"; PN2_FPS_FINE_CELLS=$f python tools/bench_fps.py --uniform 2>&1 | grep "N= *\(16384\|20000\|25000\|28672\)"; done
echo "== FINE 1 rows from PPT 16"; PN2_FPS_ROWS_MIN_PPT=16 python tools/bench_fps.py --uniform 2>&1 | grep "N= *\(16384\|20000\|25000\)"
