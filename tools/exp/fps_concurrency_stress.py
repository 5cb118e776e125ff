#!/usr/bin/env python3
"""Round 6: is pn2_fps bit-exact while OTHER kernels share the chip?  FPS of the G6 clouds (2 x 1024 -> 512) and of a 16 x 4096
batch (-> 512) on a side stream, heavy MLP kernels on the main stream, result against the same call made alone."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.nn as nn

from pointnet12_amd import pointnet_util as U
from pointnet12_amd import synthetic as syn


def main(trials=300):
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(ROOT, "tests", "golden", "g6_nets.npz"))
    small = torch.from_numpy(np.ascontiguousarray(g["points"][:, :3, :].transpose(0, 2, 1))).to(dev)
    big_np, _ = syn.kitti_batch(0, 16, 4096)
    big = torch.from_numpy(np.ascontiguousarray(big_np[:, :3, :].transpose(0, 2, 1))).to(dev)
    convs = nn.ModuleList([nn.Conv2d(a, b, 1) for a, b in ((64, 96), (96, 128))]).to(dev)
    bns = nn.ModuleList([nn.BatchNorm2d(b) for b in (96, 128)]).to(dev)
    rows = torch.randn(1 << 19, 64, device=dev)
    side = torch.cuda.Stream(device=dev)
    for name, xyz, npoint in (("2 x 1024 -> 512", small, 512), ("16 x 4096 -> 512", big, 512)):
        B, N = xyz.shape[0], xyz.shape[1]
        gen = torch.Generator().manual_seed(1)
        bad_alone = bad_conc = 0
        first = None
        for tr in range(trials):
            start = torch.randint(0, N, (B,), generator=gen).to(dev)
            ref = U.farthest_point_sample(xyz, npoint, start).clone()
            torch.cuda.synchronize()
            again = U.farthest_point_sample(xyz, npoint, start).clone()
            torch.cuda.synchronize()
            bad_alone += int(not torch.equal(ref, again))
            side.wait_stream(torch.cuda.current_stream())
            with torch.no_grad():
                U.shared_mlp(rows, 64, convs, bns, 128, True)       # main stream: two long GEMM launches + epilogues
            with torch.cuda.stream(side):
                conc = U.farthest_point_sample(xyz, npoint, start).clone()
            with torch.no_grad():
                U.shared_mlp(rows, 64, convs, bns, 128, True)
            torch.cuda.synchronize()
            if not torch.equal(ref, conc):
                bad_conc += 1
                if first is None:
                    d = (ref != conc).any(0).nonzero().flatten()
                    c0 = int(d[0])
                    first = (tr, c0, ref[:, max(0, c0 - 2):c0 + 4].tolist(), conc[:, max(0, c0 - 2):c0 + 4].tolist())
        print("%s: %d trials, differs alone %d, differs beside the MLP kernels %d; first: %s" % (name, trials, bad_alone, bad_conc, first))


if __name__ == "__main__":
    main()
