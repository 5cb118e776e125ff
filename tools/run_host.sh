O=gpurun_out/host; mkdir -p $O
for w in ssg msg; do
  python3 tools/host_times.py --workload $w > $O/$w.json 2> $O/$w.err
  for b in 4 1; do
    python3 bench.py --workload $w --batch $b --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > $O/${w}_b$b.json 2>> $O/$w.err
  done
done
tail -2 $O/ssg.err $O/msg.err; for f in $O/*.json; do python3 -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'])"; done
