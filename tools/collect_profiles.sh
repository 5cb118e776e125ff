# Collect the judged evidence of a state of the tree on the GPU box:  bash tools/collect_profiles.sh <tag>
# Writes gpurun_out/<tag>/*; copy what should be kept into profiles/ (rNN_<tag>_*).
set -u
TAG=${1:-vx}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
python3 bench.py > $O/bench_msg.json 2> $O/bench_msg.err
python3 bench.py --workload ssg > $O/bench_ssg.json 2> $O/bench_ssg.err
python3 bench.py --workload sa > $O/bench_sa.json 2> $O/bench_sa.err
python3 bench.py --workload msg --points 65536 --batch 8 --npoint-scale 16 --steps 5 --warmup 2 > $O/cfg5_msg.json 2> $O/cfg5_msg.err
python3 bench.py --workload ssg --points 65536 --batch 8 --steps 10 --warmup 3 > $O/cfg5_ssg.json 2> $O/cfg5_ssg.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_msg -o t -- python3 $R/bench.py --no-cpu-baseline > /dev/null 2> $O/prof_msg.err
PN2_MSG_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_msg_serial -o t -- python3 $R/bench.py --no-graph --no-cpu-baseline --no-roofline > /dev/null 2> $O/prof_msg_serial.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ssg -o t -- python3 $R/bench.py --workload ssg --no-cpu-baseline > /dev/null 2> $O/prof_ssg.err
cd $R
for d in prof_msg prof_msg_serial prof_ssg; do
  f=$(find $O/$d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv
  rm -rf $O/$d
done
python3 tools/bench_kernels.py all > $O/kernel_microbench.txt 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o t -- python3 bench.py --no-graph --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2> $O/pmc_$c.err
done
python3 tools/pmc_traffic.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) 5 > $O/pmc_traffic_msg.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o t -- python3 bench.py --workload ssg --no-graph --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2> $O/pmc_ssg_$c.err
done
python3 tools/pmc_traffic.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) 5 > $O/pmc_traffic_ssg.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
python3 tools/bench_infer.py > $O/infer_single_cloud.jsonl 2> $O/infer.err
python3 tools/bench_fps.py > $O/fps_probe.txt 2> $O/fps.err

python3 tools/bench_ball.py > $O/ball_query_cfg5.txt 2> $O/ball.err
python3 -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/tests_gpu.txt
ls -la $O
