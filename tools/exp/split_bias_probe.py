#!/usr/bin/env python3
"""Round 6: is the error of the bf16x3 split products DIRECTIONAL?  split_bwd_res_kernel (96 x 64) on fixed operands against fp64:
mean signed error of dX over the elements with dX > 0 and with dX < 0, in units of mean |dX| (a round-to-nearest pipe gives ~0 for
both; truncation toward zero gives opposite signs; a floor gives the SAME sign -- the one that survives a sum over rows)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch

from pointnet12_amd import _lib
import test_mlp_gpu as T


def main():
    dev = torch.device("cuda:0")
    lib, st = _lib.load(), torch.cuda.current_stream().cuda_stream
    for split in (1, 0):
        _lib.set_option("PN2_SPLIT", split)
        for log2P in (18, 21):
            P, co, ci = 1 << log2P, 96, 64
            c, ref = T._fixed_layer_case(dev, P, co, ci, 0, 5 + log2P)
            dX = torch.empty(P, c["ldp"], device=dev)
            red = torch.zeros(8 * 2 * ci, device=dev, dtype=torch.float64)
            dW = torch.zeros(co, ci, device=dev)
            rc = lib.pn2_conv1x1_bwd(*c["dz_args"], c["Y"].data_ptr(), c["ldc"], c["coef"].data_ptr(), c["W"].data_ptr(), ci, c["Yp"].data_ptr(), c["ldp"],
                                     c["affp"].data_ptr(), dX.data_ptr(), c["ldp"], red.data_ptr(), dW.data_ptr(), ci, P, co, ci, None, st)
            assert rc == 0
            torch.cuda.synchronize()
            r = ref["dX"]
            err = dX[:, :ci].double() - r
            scale = float(r.abs().mean())
            pos, neg = r > 0, r < 0
            print("PN2_SPLIT=%d 2^%d rows: mean err | dX > 0: %+.3e   mean err | dX < 0: %+.3e   rms err %.3e   (units of mean |dX|)" %
                  (split, log2P, float(err[pos].mean()) / scale, float(err[neg].mean()) / scale, float(err.pow(2).mean().sqrt()) / scale))
    _lib.set_option("PN2_SPLIT", 1)


if __name__ == "__main__":
    main()
