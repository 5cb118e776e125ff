export TMPDIR=/tmp
bash tools/exp/ab_step.sh gpurun_out/r5z/ab.txt "PN2_BENCH_FORK=top" "PN2_BENCH_FORK=sa1" "PN2_BENCH_FORK=sa2" "PN2_BENCH_FORK=loss" "PN2_BENCH_FORK=sa2_bwd" "PN2_BENCH_FORK=sa1_bwd" > /dev/null
sort gpurun_out/r5z/ab.txt
