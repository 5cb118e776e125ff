cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -m pytest $R/tests/test_modules_gpu.py $R/tests/test_parity_fullsize_gpu.py -x -q 2>&1 | tail -3
PN2_MSG_STREAMS=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/aten_tr -o t -- python3 $R/bench.py --no-graph --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2> /tmp/aten.err
f=$(find /tmp/aten_tr -name "*kernel_trace.csv" | head -1)
python $R/tools/exp/aten_list.py $f
