#!/usr/bin/env python3
"""Round 6: the values of one parameter's gradient over several runs, with and without the output-free pooled path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench
from pointnet12_amd import _lib
from pointnet12_amd import synthetic as syn
from pointnet12_amd.loss import nll_loss


def main():
    which = sys.argv[1:] or ["sa2.bn_blocks.1.2.bias", "sa2.bn_blocks.1.2.weight", "sa2.bn_blocks.0.2.bias"]
    dev = torch.device("cuda:0")
    net = bench.build_net("msg", dev)
    pts_np, lab_np = syn.kitti_batch(0, 16, 4096)
    pts, labels = torch.from_numpy(pts_np).to(dev), torch.from_numpy(lab_np).to(dev)
    params = dict(net.named_parameters())
    for cf in (2, 0):
        _lib.load()
        _lib.set_option("PN2_POOL_CF", cf)
        for r in range(3):
            torch.manual_seed(123)
            net.zero_grad(set_to_none=True)
            lp = net(pts)
            nll_loss(lp.reshape(-1, lp.shape[-1]), labels.reshape(-1)).backward()
            torch.cuda.synchronize()
            for n in which:
                g = params[n].grad
                print("POOL_CF=%d run %d %-28s max |g| %.3e  first six %s" % (cf, r, n, float(g.abs().max()), " ".join("% .3e" % v for v in g[:6].tolist())))
    _lib.set_option("PN2_POOL_CF", 2)


if __name__ == "__main__":
    main()
