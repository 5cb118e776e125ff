export TMPDIR=/tmp
mkdir -p gpurun_out/r5w
timeout 300 python3 tools/exp/queue_cadence.py > gpurun_out/r5w/cadence.txt 2>&1
cat gpurun_out/r5w/cadence.txt
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/r5w/prof -o t -- python3 /root/repo/tools/exp/queue_cadence.py > /dev/null 2>&1 )
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('/root/repo/gpurun_out/r5w/prof/t_kernel_trace.csv')))
for r in rows: r['s']=int(r['Start_Timestamp']); r['e']=int(r['End_Timestamp'])
rows.sort(key=lambda r:r['s'])
t0=rows[-60]['s']
for r in rows[-60:]:
    print(f"{(r['s']-t0)/1e3:9.1f} {(r['e']-r['s'])/1e3:7.1f} q{r['Queue_Id']} {r['Kernel_Name'][:50]}")
PY
