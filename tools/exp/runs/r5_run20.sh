export TMPDIR=/tmp
R=$(pwd)
O=$R/gpurun_out/r5x
mkdir -p $O
run() { # name, env, args
  name=$1; shift; envs=$1; shift
  ( cd /tmp && env $envs rocprofv3 --kernel-trace --output-format csv -d $O/prof_$name -o t -- python3 $R/bench.py --workload msg --no-cpu-baseline --no-roofline --steps 12 --warmup 4 "$@" > $O/bench_$name.json 2> $O/prof_$name.err )
  python3 tools/step_timeline.py $O/prof_$name/t_kernel_trace.csv --dump 1 > $O/timeline_$name.txt 2>&1
  echo "== $name"; head -10 $O/timeline_$name.txt; grep -A40 "kernels of step" $O/timeline_$name.txt | cut -c1-110
}
run noprefetch "A=1" --no-prefetch
run forktop "PN2_GEO_FORK_LATE=0"
