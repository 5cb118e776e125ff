# Collect the judged evidence of a state of the tree on the GPU box:  bash tools/collect_profiles.sh <tag> [quick]
# Writes gpurun_out/<tag>/*; copy what should be kept into profiles/ (rNN_<tag>_*).
# Every rocprofv3 command puts python3 itself behind `--` (no wrapper hop: the profiler's preloaded library has already
# initialised the GPU), and counters are collected in passes of their own (--pmc with --kernel-trace only).
set -u
TAG=${1:-vx}
QUICK=${2:-}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
CFG5_MSG="--workload msg --points 65536 --batch 8 --npoint-scale 16"
CFG5_SSG="--workload ssg --points 65536 --batch 8"

pmc() {   # pmc <name> <bench args...>: FETCH_SIZE / WRITE_SIZE passes of 3 + 2 eager steps -> $O/pmc_traffic_<name>.json
  local name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o t -- python3 bench.py "$@" --no-graph --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2> $O/pmc_${name}_$c.err
  done
  python3 tools/pmc_traffic.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) 5 > $O/pmc_traffic_$name.json
  rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
}
stats() { # stats <name> <env or empty> <bench args...>: rocprofv3 --kernel-trace --stats -> $O/prof_<name>_kernel_stats.csv
  local name=$1; shift
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o t -- python3 $R/bench.py "$@" --no-cpu-baseline > /dev/null 2> $O/prof_$name.err )
  f=$(find $O/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/prof_${name}_kernel_stats.csv
  rm -rf $O/prof_$name
}

# 1. counters first: the bench lines below then quote digest-matched traffic (bench.py reads profiles/r*_pmc_traffic_<key>.json)
pmc msg
pmc ssg --workload ssg
pmc sa --workload sa
pmc ssg_n65536 $CFG5_SSG
pmc msg_n65536 $CFG5_MSG
mkdir -p $R/profiles
for k in msg ssg sa ssg_n65536 msg_n65536; do cp $O/pmc_traffic_$k.json $R/profiles/${TAG}_pmc_traffic_$k.json; done

# 2. the bench lines (default command first: exactly what the driver runs)
python3 bench.py > $O/bench_msg.json 2> $O/bench_msg.err
python3 bench.py --workload ssg > $O/bench_ssg.json 2> $O/bench_ssg.err
python3 bench.py --workload sa > $O/bench_sa.json 2> $O/bench_sa.err
python3 bench.py $CFG5_MSG --steps 5 --warmup 2 > $O/cfg5_msg.json 2> $O/cfg5_msg.err
python3 bench.py $CFG5_SSG --steps 10 --warmup 3 > $O/cfg5_ssg.json 2> $O/cfg5_ssg.err

# 3. rocprofv3 --kernel-trace --stats of the same commands (graph replay; serial launches; the stand-alone BatchNorm launches for A/B)
stats msg
PN2_MSG_STREAMS=0 stats msg_serial --no-graph --no-roofline
stats ssg --workload ssg
stats ssg_serial --workload ssg --no-graph --no-roofline
PN2_LAZY_BN=0 stats ssg_serial_standalone_bn --workload ssg --no-graph --no-roofline
PN2_LAZY_BN=0 PN2_MSG_STREAMS=0 stats msg_serial_standalone_bn --no-graph --no-roofline
stats sa --workload sa
stats cfg5_ssg $CFG5_SSG --steps 10 --warmup 3
stats cfg5_msg $CFG5_MSG --steps 5 --warmup 2

# 4. A/B lines of the round's switches (same box, back to back, twice)
: > $O/ab_switches.txt
for rep in 1 2; do
  for v in "PN2_SPLIT=1" "PN2_POOL_CF=0" "PN2_POOL_CF=1" "PN2_POOL_CF=2" "PN2_SPLIT_WG2=0" "PN2_SPLIT_WG2=1" "PN2_FPS_PIECE=128" "PN2_FUSE_FIRST=0" "PN2_FUSE_FIRST=1" "PN2_SPLIT=0" "PN2_SPLIT_RES=0" \
           "PN2_SPLIT_NARROW=0" "PN2_SPLIT_WGRAD=0" "PN2_SPLIT_K256=0" "PN2_LAZY_BN=0" "PN2_WIDE_POOL=0" "PN2_BWD_PAIR=0"; do
    for w in msg ssg; do
      env $v python3 bench.py --workload $w --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], sys.argv[2], d['ms_per_step'])" $v $w >> $O/ab_switches.txt
    done
  done
done

# 5. the captured step as the chip ran it: rocprofv3 kernel trace -> per-step wall time, queues, time attributed per kernel family
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/trace_msg -o t -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 > /dev/null 2> $O/trace_msg.err )
python3 tools/step_timeline.py $(find $O/trace_msg -name "*kernel_trace.csv" | head -1) --dump 1 > $O/step_timeline_msg.txt 2>&1
rm -rf $O/trace_msg
# 6. in-kernel stamps of the bf16-split kernels (a STAMP build beside the product library: tools/exp/build_variant.sh stamp "" 1)
if [ -f pointnet12_amd/libpn2_hip_stamp.so ]; then
  PN2_LIB_PATH=pointnet12_amd/libpn2_hip_stamp.so python3 tools/stamp_wide.py > $O/stamp_split_nt.txt 2>&1
  PN2_SPLIT_RES=2 PN2_LIB_PATH=pointnet12_amd/libpn2_hip_stamp.so python3 tools/stamp_res.py > $O/stamp_split_bwd_res.txt 2>&1
fi

# SQ / GRBM counters of the fused backward kernels alone (clock held, matrix-pipe busy share, VALU : MFMA, LDS): tools/pmc_kernels.sh
( bash tools/pmc_kernels.sh bwd 1048576,524288 $O/sq_bwd > /dev/null 2>&1; echo "=== bwd"; cat $O/sq_bwd/summary.txt
  bash tools/pmc_kernels.sh bwdcf 1048576,524288 $O/sq_bwdcf > /dev/null 2>&1; echo "=== bwdcf"; cat $O/sq_bwdcf/summary.txt ) > $O/sq_counters_split.txt 2>&1
rm -rf $O/sq_bwd $O/sq_bwdcf
python3 tools/exp/split_bias_probe.py > $O/split_bias_probe.txt 2>&1
python3 tools/exp/nostore_probe.py > $O/nostore_probe.txt 2>&1
tools/exp/mfma_peak > $O/mfma_peak.txt 2>&1
[ -x tools/exp/mfma_shape ] && tools/exp/mfma_shape > $O/mfma_shape.txt 2>&1
if [ -z "$QUICK" ]; then
  python3 tools/bench_kernels.py all > $O/kernel_microbench.txt 2>/dev/null
  python3 tools/bench_infer.py > $O/infer_single_cloud.jsonl 2> $O/infer.err
  python3 tools/bench_fps.py > $O/fps_probe.txt 2> $O/fps.err
  python3 tools/bench_ball.py > $O/ball_query_cfg5.txt 2> $O/ball.err
  python3 -m pytest tests -m gpu -q > $O/tests_gpu_full.txt 2>&1; (grep "^E  " $O/tests_gpu_full.txt | head -20; tail -5 $O/tests_gpu_full.txt) > $O/tests_gpu.txt
  cp gpurun_out/rccl_same_device.txt $O/rccl_same_device.txt 2>/dev/null
  cp gpurun_out/parity_stages.json $O/parity_stages.json 2>/dev/null
  cp gpurun_out/parity_fullsize.json $O/parity_fullsize.json 2>/dev/null
fi
ls -la $O
