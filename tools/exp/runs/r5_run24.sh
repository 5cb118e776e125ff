export TMPDIR=/tmp
O=gpurun_out/r5ab
mkdir -p $O
PN2_LIB_PATH=pointnet12_amd/libpn2_hip_stamp.so timeout 300 python tools/stamp_res.py 2>&1 | grep -v "^fwd\|transform" > $O/stamp.txt
cat $O/stamp.txt
