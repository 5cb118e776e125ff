python -m pytest tests/test_mlp_gpu.py tests/test_modules_gpu.py -x -q 2>&1 | tail -2
python tools/bench_kernels.py dgrad --only 2048,8192 2>&1 | grep -v amdgpu
for m in msg ssg; do
python bench.py --workload $m --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m new', d['ms_per_step'])"
PN2_FEWROW_MAX_TILES=0 PN2_FEWROW_MAX_TILES_DGRAD=0 python bench.py --workload $m --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m old', d['ms_per_step'])"
done
