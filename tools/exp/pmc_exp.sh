# GPU box: timing of every case of an experiment binary, then SQ / GRBM counters of selected cases (separate --pmc passes).
#   bash tools/exp/pmc_exp.sh <binary> <out dir> <case ids...>
export TMPDIR=/tmp
BIN=$1; OUT=$2; shift 2
mkdir -p $OUT
$BIN > $OUT/timing.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  for c in "$@"; do
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p${i}_c$c -o t -- $BIN $c > $OUT/p${i}_c$c.out 2> $OUT/p${i}_c$c.err
  done
done
python3 tools/exp/pmc_exp.py $OUT > $OUT/summary.txt
cat $OUT/timing.txt $OUT/summary.txt
