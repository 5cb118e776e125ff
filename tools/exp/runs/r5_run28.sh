export TMPDIR=/tmp
O=gpurun_out/r5af
mkdir -p $O
for v in stamp st_nostore; do
echo "== $v"
PN2_SPLIT_RES=2 PN2_LIB_PATH=pointnet12_amd/libpn2_hip_$v.so timeout 300 python tools/stamp_res.py 2>&1 | grep -v "^fwd\|transform" > $O/stamp_$v.txt
grep -A9 "^bwd (1048576, 128, 96" $O/stamp_$v.txt | cut -c1-200
done
