export TMPDIR=/tmp
R=$(pwd)
O=$R/gpurun_out/r5bi
mkdir -p $O
: > $O/ab.txt
for arm in "A=0" "PN2_GEO_SEPARATE=1" "PN2_GEO_SEPARATE=1" "A=0" "A=0" "PN2_GEO_SEPARATE=1"; do
  for w in msg ssg; do
    env $arm python3 bench.py --workload $w --no-cpu-baseline --no-roofline 2>$O/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], '|', sys.argv[2], d['ms_per_step'])" "$arm" $w >> $O/ab.txt || tail -5 $O/err.txt
  done
done
sort $O/ab.txt
( cd /tmp && PN2_GEO_SEPARATE=1 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 12 --warmup 4 > /dev/null 2> $O/trace.err )
python3 tools/step_timeline.py $(find $O/trace -name "*kernel_trace.csv" | head -1) --dump 1 > $O/timeline.txt 2>&1
head -12 $O/timeline.txt
grep -n "fps_kernel<512" -B6 -A12 $O/timeline.txt | tail -22 | cut -c1-110
