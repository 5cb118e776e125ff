#!/usr/bin/env python3
"""Round 6: the reverse of fps_concurrency_stress2.py -- are the MLP kernels' outputs bit-stable while pn2_fps runs beside them?

The pooled bf16-split forward (output, pooling records), the dense split forward and the fused backward (dX) are launched with fixed
operands on the main stream while a side stream loops FPS launches; every result is compared bit for bit with the launch alone."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from pointnet12_amd import _lib
from pointnet12_amd import pointnet_util as U
from pointnet12_amd import synthetic as syn
from pointnet12_amd._lib import ptr as p


def main(trials=100):
    dev = torch.device("cuda:0")
    lib = _lib.load()
    big_np, _ = syn.kitti_batch(0, 16, 4096)
    xyz = torch.from_numpy(np.ascontiguousarray(big_np[:, :3, :].transpose(0, 2, 1))).to(dev)
    start = torch.zeros(16, dtype=torch.int64, device=dev)
    P = 1 << 19
    g = torch.Generator(device=dev).manual_seed(0)
    X64, X96 = torch.randn(P, 64, device=dev, generator=g), torch.randn(P, 96, device=dev, generator=g)
    W96, W128 = torch.randn(96, 64, device=dev, generator=g), torch.randn(128, 96, device=dev, generator=g)
    b96, b128 = torch.randn(96, device=dev, generator=g), torch.randn(128, device=dev, generator=g)
    aff64 = torch.zeros(4 * 64, device=dev); aff64[64:128] = 1; aff64[192:] = 1
    aff96 = torch.zeros(4 * 96, device=dev); aff96[96:192] = 1; aff96[288:] = 1
    main_s = torch.cuda.current_stream().cuda_stream

    def fwd_pool():
        Y = torch.empty(P, 128, device=dev)
        ws = torch.zeros(2 * (P // 128) * 128, device=dev)
        st = torch.zeros(8 * 2 * 128, device=dev, dtype=torch.float64)
        assert lib.pn2_conv1x1_fwd_pool(p(X96), 96, p(aff96), p(W128), 96, p(b128), p(Y), 128, P, 96, 128, p(st), 128, p(b128), p(ws), None, main_s) == 0
        return Y, ws

    def fwd_dense():
        Y = torch.empty(P, 96, device=dev)
        st = torch.zeros(8 * 2 * 96, device=dev, dtype=torch.float64)
        assert lib.pn2_conv1x1_fwd(p(X64), 64, p(aff64), p(W96), 64, p(b96), p(Y), 96, P, 64, 96, p(st), None, None, main_s) == 0
        return (Y,)

    side = torch.cuda.Stream(device=dev)
    for name, fn in (("pooled split forward 96 -> 128 (Y, pooling records)", fwd_pool), ("split forward 64 -> 96 with BN input (Y)", fwd_dense)):
        ref = [t.clone() for t in fn()]
        torch.cuda.synchronize()
        bad = 0
        for tr in range(trials):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    U.farthest_point_sample(xyz, 512, start)
            out = fn()
            torch.cuda.synchronize()
            bad += int(not all(torch.equal(a, b) for a, b in zip(ref, out)))
        print("%-60s: %d of %d launches beside pn2_fps differ from the launch alone" % (name, bad, trials))


if __name__ == "__main__":
    main()
