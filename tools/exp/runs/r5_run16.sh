export TMPDIR=/tmp
R=$(pwd)
O=$R/gpurun_out/r5t
mkdir -p $O
for v in o0 main main; do
  if [ $v = main ]; then unset PN2_LIB_PATH; else export PN2_LIB_PATH=$R/pointnet12_amd/libpn2_hip_$v.so; fi
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o t -- python3 $R/bench.py --workload msg --no-cpu-baseline --no-roofline --steps 20 --warmup 5 > $O/bench_$v.json 2> $O/prof_$v.err )
  f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1)
  python3 - "$f" > $O/split_$v.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms per 25 steps', tot/1e6)
for r in rows:
    n=r['Name']
    if 'split_' in n:
        print(f"{float(r['AverageNs'])/1e3:9.1f} us x{r['Calls']:>5}  {float(r['TotalDurationNs'])/1e6:8.2f} ms  {n[n.find('split_'):n.find('split_')+90]}")
PY
  cat $O/split_$v.txt
done
