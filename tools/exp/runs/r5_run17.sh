export TMPDIR=/tmp
O=gpurun_out/r5u
mkdir -p $O
timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_modules_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
tail -2 $O/tests.txt
bash tools/exp/ab_step.sh $O/ab.txt "PN2_LIB_PATH=pointnet12_amd/libpn2_hip_o0.so" "-" "PN2_LIB_PATH=pointnet12_amd/libpn2_hip_o0.so" "-"
