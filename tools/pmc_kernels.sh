# GPU box: SQ / GRBM counters of the GEMM microbench (tools/bench_kernels.py), one rocprofv3 --pmc pass per counter set.
#   bash tools/pmc_kernels.sh <which: fwd|bwd|dgrad|wgrad> <P list> <out dir>
# GRBM_GUI_ACTIVE / 8 / kernel time = effective clock (MI355X_MICROARCH.md, DVFS give-back); SQ_VALU_MFMA_BUSY_CYCLES
# / (GRBM_GUI_ACTIVE / 8) / (4 SIMDs x CUs) = matrix-pipe utilisation.
export TMPDIR=/tmp
W=$1; PL=$2; OUT=$3
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o t -- python3 tools/bench_kernels.py $W --only $PL --reps 5 > $OUT/p$i.out 2> $OUT/p$i.err
done
python3 tools/pmc_kernels.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
