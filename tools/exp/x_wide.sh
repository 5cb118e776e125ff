#!/bin/bash
# Ablation driver used in round 4 to price the pieces of the register-stationary kernels (mlp_wide.hip) on ONE box: the
# switches it names (-DPN2_X_*) were temporary edits of the working tree (skip the staging / the requests / the dW flush /
# the chunk barrier / the prevY reads of the masked epilogue) and are NOT in the sources; the numbers are in HISTORY.md
# section 4 and profiles/r04_ablation_wide.txt.  Kept as the recipe: patch, `make XFLAGS=...`, run the four launches.
cd "$GRAFT_REPO_ROOT" || exit 1
run() {
  for sh in 262144,256,196 262144,196,128; do python tools/bench_kernels.py dgrad --shape $sh --reps 30 2>/dev/null; done
  for sh in 262144,256,196 262144,196,128; do python tools/bench_kernels.py wgrad --shape $sh --reps 30 2>/dev/null; done
}
echo "== baseline"; run
for X in "$@"; do
  touch pointnet12_amd/csrc/mlp_wide.hip
  make -C pointnet12_amd/csrc XFLAGS="$X" -j16 >/dev/null 2>&1 || echo BUILD FAILED
  echo "== $X"; run
done
