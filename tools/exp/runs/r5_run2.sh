# GPU box, round 5 run 2: ring forward with float4-unit LDS addressing; MFMA order A/B; in-kernel stamps
export TMPDIR=/tmp
O=gpurun_out/r5b
mkdir -p $O
timeout 600 python -m pytest tests/test_mlp_gpu.py -x -q -m gpu > $O/tests_mlp.txt 2>&1
tail -2 $O/tests_mlp.txt
for rep in 1 2; do
  PN2_RING=0 timeout 300 python tools/bench_kernels.py fwd --only 262144,131072 > $O/fwd_ring0_$rep.txt 2>&1
  PN2_RING=1 PN2_RING_ORD=0 timeout 300 python tools/bench_kernels.py fwd --only 262144,131072 > $O/fwd_ring1_ord0_$rep.txt 2>&1
  PN2_RING=1 PN2_RING_ORD=1 timeout 300 python tools/bench_kernels.py fwd --only 262144,131072 > $O/fwd_ring1_ord1_$rep.txt 2>&1
done
grep -H "196, 256\|128, 196\|128, 256\|131072, 128, 128" $O/fwd_ring*.txt | sed 's/.*fwd_//'
PN2_LIB_PATH=pointnet12_amd/libpn2_hip_stamp.so PN2_RING_ORD=0 timeout 300 python tools/stamp_wide.py > $O/stamp_ord0.txt 2>&1
PN2_LIB_PATH=pointnet12_amd/libpn2_hip_stamp.so PN2_RING_ORD=1 timeout 300 python tools/stamp_wide.py > $O/stamp_ord1.txt 2>&1
cat $O/stamp_ord0.txt $O/stamp_ord1.txt
