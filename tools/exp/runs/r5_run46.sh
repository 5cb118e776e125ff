export TMPDIR=/tmp
R=$(pwd)
O=$R/gpurun_out/r5bg
mkdir -p $O
for m in ingraph twographs; do
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/prof_$m -o t -- python3 $R/tools/exp/queue_cadence2.py $m > /dev/null 2> $O/err_$m.txt )
python3 - $O/prof_$m <<'PY'
import csv,sys,glob,re
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows: r['s']=int(r['Start_Timestamp']); r['e']=int(r['End_Timestamp'])
rows.sort(key=lambda r:r['s'])
# last FPS launch and the 20 kernels around it
idx=[i for i,r in enumerate(rows) if 'fps_kernel' in r['Kernel_Name']][-1]
t0=rows[idx]['s']
print(sys.argv[1].split('_')[-1])
for r in rows[max(0,idx-3):idx+16]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name'])[:48]
    print(f"  {(r['s']-t0)/1e3:8.1f} {(r['e']-r['s'])/1e3:7.1f} q{r['Queue_Id']} {n}")
PY
done
