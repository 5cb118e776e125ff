# A/B of one environment switch on the replayed step: bash tools/exp/switch_sweep.sh VAR "v1 v2 ..." [workloads]
VAR=$1; VALS=$2; WL=${3:-"msg ssg"}
for w in $WL; do for v in $VALS; do
  env $VAR=$v python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; print('$w $VAR=$v', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
done; done
