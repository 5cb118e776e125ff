export TMPDIR=/tmp
O=gpurun_out/r5am
mkdir -p $O
PN2_LIB_PATH=pointnet12_amd/libpn2_hip_stamp.so timeout 300 python tools/stamp_wide.py > $O/stamp.txt 2>&1
grep -A10 "^wgrad" $O/stamp.txt | cut -c1-220
tail -5 $O/stamp.txt | cut -c1-300
