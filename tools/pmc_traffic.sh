# Collect HBM traffic counters for one bench step (run on the GPU box): two PMC passes, as the TCC block cannot
# hold FETCH_SIZE and WRITE_SIZE together (MI355X_MICROARCH.md, rocprofv3 PMC slots).
export TMPDIR=/tmp
R=$PWD
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -o t -- python3 bench.py --no-graph --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2> $R/gpurun_out/pmc_$c.err
done
python3 tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE/t_counter_collection.csv gpurun_out/pmc_WRITE_SIZE/t_counter_collection.csv 5 > gpurun_out/pmc_traffic.json
cat gpurun_out/pmc_traffic.json
