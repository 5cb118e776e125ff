cd pointnet12_amd/csrc
for v in "" "-DPN2_X_NOWATOMIC"; do
  make clean >/dev/null; make -s -j4 XFLAGS="$v" 2>&1 | grep -E " error"
  echo "VARIANT [$v]"; (cd ../..; python tools/bench_kernels.py wgrad 2>&1 | grep wgrad)
done
make clean >/dev/null; make -s -j4 2>&1 | grep -E " error"
