export TMPDIR=/tmp
O=gpurun_out/r5bj
mkdir -p $O
bash tools/pmc_kernels.sh fwd 262144,131072,1048576 $O/fwd > /dev/null 2>&1
bash tools/pmc_kernels.sh dgrad 262144,131072 $O/dgrad > /dev/null 2>&1
bash tools/pmc_kernels.sh wgrad 262144,131072 $O/wgrad > /dev/null 2>&1
bash tools/pmc_kernels.sh bwd 1048576,524288 $O/bwd > /dev/null 2>&1
for w in fwd dgrad wgrad bwd; do echo "=== $w"; cat $O/$w/summary.txt; rm -rf $O/$w/p1 $O/$w/p2 $O/$w/p3 $O/$w/p4; done > $O/sq_counters_split.txt
wc -l $O/sq_counters_split.txt; head -60 $O/sq_counters_split.txt | cut -c1-230
