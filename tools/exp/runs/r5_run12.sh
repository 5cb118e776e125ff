export TMPDIR=/tmp
O=gpurun_out/r5p
mkdir -p $O
timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_modules_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
tail -3 $O/tests.txt
PN2_LIB_PATH=pointnet12_amd/libpn2_hip_stamp.so timeout 300 python tools/stamp_wide.py > $O/stamp_split.txt 2>&1
grep -v "wave [1235679]" $O/stamp_split.txt
bash tools/exp/ab_step.sh $O/ab.txt "PN2_SPLIT_NARROW=0" "-"
