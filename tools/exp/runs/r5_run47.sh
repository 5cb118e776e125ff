export TMPDIR=/tmp
R=$(pwd)
O=$R/gpurun_out/r5bh
mkdir -p $O
: > $O/ab.txt
for arm in "A=0" "PN2_MSG_STREAMS=0" "PN2_MSG_MAIN_LAST=0" "A=0" "PN2_MSG_STREAMS=0" "PN2_MSG_MAIN_LAST=0"; do
  env $arm python3 bench.py --workload msg --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], '|', d['ms_per_step'])" "$arm" >> $O/ab.txt
done
sort $O/ab.txt
( cd /tmp && PN2_MSG_STREAMS=0 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 12 --warmup 4 > /dev/null 2> $O/trace.err )
python3 tools/step_timeline.py $(find $O/trace -name "*kernel_trace.csv" | head -1) --dump 1 > $O/timeline_serial.txt 2>&1
head -12 $O/timeline_serial.txt
grep -A40 "kernels of step" $O/timeline_serial.txt | grep -n "fps_kernel<512" -B6 -A14 | cut -c1-110
