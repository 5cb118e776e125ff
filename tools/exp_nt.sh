# diagnostic: stamped build + config sweep of the NT GEMM (run on the GPU box)
cd pointnet12_amd/csrc
make clean >/dev/null; make -s -j4 STAMP=1 2>&1 | grep -E " error"
cd ../..
for c in 0 3; do echo "STAMPS cfg $c"; PN2_NT_CFG=$c python tools/stamp_nt.py 2>&1 | grep -v amdgpu.ids; done
cd pointnet12_amd/csrc; make clean >/dev/null; make -s -j4 2>&1 | grep -E " error"; cd ../..
for c in 0 2 3 4; do echo "CFG $c"; PN2_NT_CFG=$c python tools/bench_kernels.py fwd 2>&1 | grep -E "96, 128\)|64, 96\)|323, 128\)|196, 256|128, 196"; PN2_NT_CFG=$c python tools/bench_kernels.py dgrad 2>&1 | grep dgrad | head -5; done
