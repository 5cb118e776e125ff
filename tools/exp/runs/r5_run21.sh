export TMPDIR=/tmp
bash tools/exp/ab_step.sh gpurun_out/r5y/ab.txt "PN2_GEO_FORK_LATE=0" "-" "PN2_GEO_FORK_LATE=0" "-" > /dev/null
sort gpurun_out/r5y/ab.txt
