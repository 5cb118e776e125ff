#!/bin/bash
# same-box A/B of the output-free pooled last layer (PN2_POOL_CF) per kernel: rocprofv3 stats of the serial eager step
export PN2_MSG_STREAMS=0
for cf in 0 1 2; do
  export PN2_POOL_CF=$cf
  echo "== POOL_CF=$cf"
  tools/exp/prof_cmd.sh r6g_cf$cf bench.py --no-graph --no-roofline --no-cpu-baseline --no-other-configs 2>&1 | grep "split_bwd_res_kernel<4, [23], true\|split_bwd_cf\|cf_finish\|cf_prep\|split_nt_kernel<96, 4, 1, 2, 1, 0, false, 128\|split_nt_kernel<64, 4, 1, 2, 1, 0, false, 64\|pool_bwd_reduce"
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/r6g_cf$cf/kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows if "at::" not in r["Name"] and "rocclr" not in r["Name"])
print("library kernels total per step: %.3f ms" % (tot/25/1e6))
PY
done
unset PN2_POOL_CF PN2_MSG_STREAMS
for cf in 0 1 2 0 1 2; do PN2_POOL_CF=$cf python3 bench.py --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('POOL_CF=$cf', d['ms_per_step'])"; done
python3 bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['kernel'], r['bound'], r['frac'], r['avg_us'])
for k in r['kernels']: print('%-80s %s %-5s %.3f %8.1f us x%d' % (k['kernel'][:80], k['pipe'], k['bound'], k['frac'], k['avg_us'], k['launches_per_step']))"
