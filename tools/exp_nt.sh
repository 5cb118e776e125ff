cd pointnet12_amd/csrc
for v in "" "-DPN2_X_NOSTORE" "-DPN2_X_NOBLOAD" "-DPN2_X_NOALOAD" "-DPN2_X_NOSTORE -DPN2_X_NOBLOAD -DPN2_X_NOALOAD"; do
  make clean >/dev/null; make -s -j4 XFLAGS="$v" 2>&1 | grep -E " error" ; 
  echo "VARIANT [$v]"; (cd ../..; for c in 0 1; do PN2_NT_CFG=$c python tools/bench_kernels.py fwd 2>&1 | grep -E "96, 128\)|64, 96\)|323, 128\)|196, 256" | sed "s/^/cfg$c /"; done)
done
