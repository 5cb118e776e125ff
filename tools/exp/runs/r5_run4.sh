# GPU box, round 5 run 4: the stage parity test with decisions counted and forced; MLP suite with the ring forward on
export TMPDIR=/tmp
O=gpurun_out/r5f
mkdir -p $O
timeout 1500 python -m pytest tests/test_parity_stages_gpu.py -x -q -m gpu > $O/tests_stages.txt 2>&1
tail -30 $O/tests_stages.txt
cp gpurun_out/parity_stages.json $O/parity_stages.json 2>/dev/null
timeout 900 python -m pytest tests/test_mlp_gpu.py -x -q -m gpu > $O/tests_ring.txt 2>&1
tail -3 $O/tests_ring.txt
