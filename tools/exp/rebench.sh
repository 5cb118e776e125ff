#!/bin/bash
# the five bench lines of tools/collect_profiles.sh only (profiles/r04_pmc_traffic_* of the same sources already installed)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04b; mkdir -p $O
python3 bench.py > $O/bench_msg.json 2> $O/bench_msg.err
python3 bench.py --workload ssg > $O/bench_ssg.json 2> $O/bench_ssg.err
python3 bench.py --workload sa > $O/bench_sa.json 2> $O/bench_sa.err
python3 bench.py --workload msg --points 65536 --batch 8 --npoint-scale 16 --steps 5 --warmup 2 > $O/cfg5_msg.json 2> $O/cfg5_msg.err
python3 bench.py --workload ssg --points 65536 --batch 8 --steps 10 --warmup 3 > $O/cfg5_ssg.json 2> $O/cfg5_ssg.err
python -m pytest tests/test_mlp_gpu.py -q -k closed_form 2>&1 | tail -2
for rep in 1 2; do for v in PN2_GEO_FORK_LATE=0 PN2_GEO_FORK_LATE=1; do for w in msg ssg; do
  env $v python3 bench.py --workload $w --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], sys.argv[2], d['ms_per_step'])" $v $w >> $O/ab_fork.txt
done; done; done
cat $O/ab_fork.txt
