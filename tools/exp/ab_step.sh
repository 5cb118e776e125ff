# GPU box: same-box A/B of option settings at the step level.   bash tools/exp/ab_step.sh OUT "VAR=a VAR2=b" "VAR=c" ...
# Each argument after OUT is one arm (a space-separated list of PN2_*=value settings; "-" = defaults); two repetitions, msg + ssg.
O=$1; shift
mkdir -p $(dirname $O)
: > $O
for rep in 1 2; do for arm in "$@"; do for w in msg ssg; do
  if [ "$arm" = "-" ]; then envs=""; else envs="$arm"; fi
  env $envs python3 bench.py --workload $w --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], '|', sys.argv[2], d['ms_per_step'])" "$arm" $w >> $O
done; done; done
cat $O
