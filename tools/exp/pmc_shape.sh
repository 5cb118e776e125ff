# GPU box: counters of ONE GEMM shape through tools/bench_kernels.py (separate --pmc passes).
#   bash tools/exp/pmc_shape.sh <fwd|dgrad|wgrad> <P,K,N> <out dir>
export TMPDIR=/tmp
W=$1; SH=$2; OUT=$3
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" \
           "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p${i}_c0 -o t -- python3 tools/bench_kernels.py $W --shape $SH --reps 5 > $OUT/p$i.out 2> $OUT/p$i.err
done
python3 - $OUT <<'PY'
import collections, csv, glob, os, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "p*_c0", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3)
for k in agg:
    us = sorted(dur[k])[len(dur[k]) // 2]
    if us < 8:
        continue
    c = {n: sum(v) / len(v) for n, v in agg[k].items()}
    print("%s  %.1f us" % (k, us))
    for n in sorted(c):
        print("    %-40s %.4g" % (n, c[n]))
PY
