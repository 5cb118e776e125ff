export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/tl; mkdir -p $O
cd /tmp
for w in ssg msg; do
  rocprofv3 --kernel-trace --output-format csv -d $O/p_$w -o t -- python3 $R/bench.py --workload $w --no-cpu-baseline --no-roofline --steps 6 --warmup 3 > $O/$w.json 2> $O/$w.err
  f=$(find $O/p_$w -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/trace_timeline.py $f $O/timeline_$w.txt
  rm -rf $O/p_$w
done
head -5 $O/timeline_ssg.txt
