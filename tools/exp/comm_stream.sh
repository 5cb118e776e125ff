python -m pytest tests/test_parallel_gpu.py -x -q 2>&1 | tail -5
export PN2_FORCE_COLLECTIVES=1
for w in msg ssg; do for flag in "" "--no-comm-stream"; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29540 bench.py --gpus 1 --steps 30 --warmup 5 --workload $w --no-cpu-baseline --no-roofline $flag 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('$w flag=[$flag]', d['ms_per_step'], d['allreduce_ms'], d['config']['allreduce_stream'])"
done; done
