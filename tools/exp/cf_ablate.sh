#!/bin/bash
# the CF kernel with one piece dropped at a time (variants built by tools/exp/build_variant.sh cf_<X> -DPN2_X_CF_<X>)
for v in "" cf_NOT cf_NODXM cf_NOSTAGE cf_NOEPI cf_NOFETCH; do
  if [ -n "$v" ]; then export PN2_LIB_PATH=$PWD/pointnet12_amd/libpn2_hip_$v.so; else unset PN2_LIB_PATH; fi
  echo "== ${v:-baseline}"
  tools/exp/prof_cmd.sh r6d_${v:-base} tools/bench_kernels.py bwdcf 2>&1 | grep "split_bwd_cf\|cf_finish\|cf_prep"
done
