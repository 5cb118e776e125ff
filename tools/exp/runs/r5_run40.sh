export TMPDIR=/tmp
O=gpurun_out/r5ax
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/all.txt 2>&1
tail -3 $O/all.txt; grep "^E   \|^FAILED" $O/all.txt | head
bash tools/exp/ab_step.sh $O/ab.txt "-" "PN2_SPLIT_RES_MIN_TILES_128=4096" "-" "PN2_SPLIT_RES_MIN_TILES_128=4096" > /dev/null
sort $O/ab.txt
