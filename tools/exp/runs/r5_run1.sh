# GPU box, round 5 run 1: the LDS-DMA ring forward -- correctness (the MLP suite), same-box A/B against the register-staged
# kernel (PN2_RING=0), SQ / GRBM counters of both.   bash tools/exp/r5_run1.sh
export TMPDIR=/tmp
O=gpurun_out/r5a
mkdir -p $O
timeout 900 python -m pytest tests/test_mlp_gpu.py -x -q -m gpu > $O/tests_mlp.txt 2>&1
tail -5 $O/tests_mlp.txt
for rep in 1 2; do
  for ring in 0 1; do
    PN2_RING=$ring timeout 300 python tools/bench_kernels.py fwd --only 262144,131072 > $O/fwd_ring${ring}_$rep.txt 2>&1
  done
done
grep -h "196, 256\|128, 196\|128, 256\|131072, 128, 128" $O/fwd_ring*.txt
PN2_RING=1 timeout 600 bash tools/pmc_kernels.sh fwd 262144 $O/pmc_ring1 > /dev/null 2>&1
PN2_RING=0 timeout 600 bash tools/pmc_kernels.sh fwd 262144 $O/pmc_ring0 > /dev/null 2>&1
grep -A1 "ring_fwd\|regw_nt" $O/pmc_ring1/summary.txt $O/pmc_ring0/summary.txt
rm -rf $O/pmc_ring*/p*/ 
