# HBM traffic per step per kernel family of the current tree (two rocprofv3 --pmc passes per workload, kernel-trace only):
#   bash tools/collect_pmc.sh <tag>   ->  gpurun_out/<tag>/pmc_traffic_{msg,ssg}.json   (copy to profiles/rNN_pmc_traffic_*.json)
set -u
TAG=${1:-pmc}
export TMPDIR=/tmp
O=$PWD/gpurun_out/$TAG
mkdir -p $O
for w in msg ssg; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o t -- python3 bench.py --workload $w --no-graph --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2> $O/pmc_${w}_$c.err
  done
  python3 tools/pmc_traffic.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) 5 > $O/pmc_traffic_$w.json
  rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
done
