export TMPDIR=/tmp
O=gpurun_out/r5ay
mkdir -p $O
: > $O/ab.txt
for arm in 4096 1024 1024 4096 4096 1024 1024 4096 4096 1024; do
  for w in ssg msg; do
    PN2_SPLIT_RES_MIN_TILES_128=$arm python3 bench.py --workload $w --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], sys.argv[2], d['ms_per_step'])" $arm $w >> $O/ab.txt
  done
done
cat $O/ab.txt
