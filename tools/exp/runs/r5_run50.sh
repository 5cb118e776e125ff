export TMPDIR=/tmp
O=gpurun_out/r5bk
mkdir -p $O
: > $O/ab.txt
for arm in "A=0" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=16" "GPU_MAX_HW_QUEUES=2" "A=0" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=16" "GPU_MAX_HW_QUEUES=2"; do
  for w in msg ssg; do
    env $arm python3 bench.py --workload $w --no-cpu-baseline --no-roofline 2>$O/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], '|', sys.argv[2], d['ms_per_step'])" "$arm" $w >> $O/ab.txt || tail -3 $O/err.txt
  done
done
sort $O/ab.txt
