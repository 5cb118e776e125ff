export TMPDIR=/tmp
O=gpurun_out/r5ad
mkdir -p $O
bash tools/exp/ab_step.sh $O/ab.txt "PN2_SPLIT_RES=0" "-" "PN2_SPLIT_RES=2" "PN2_SPLIT_RES=0" "-" "PN2_SPLIT_RES=2" > /dev/null
sort $O/ab.txt
