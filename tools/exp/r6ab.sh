#!/bin/bash
ok=0; for i in $(seq 1 12); do
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600+i)) bench.py --gpus 1 --steps 3 --warmup 2 --workload ssg --no-cpu-baseline --no-roofline > gpurun_out/r6ab_$i.out 2> gpurun_out/r6ab_$i.err
  rc=$?; echo "run $i rc=$rc"; [ $rc = 0 ] && ok=$((ok+1)) && rm -f gpurun_out/r6ab_$i.out gpurun_out/r6ab_$i.err
done
echo "ok $ok of 12"
