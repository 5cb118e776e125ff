#!/usr/bin/env python3
"""How far apart are the loss sequences of the eager step and the two captured schedules (tests/test_modules_gpu.py:
test_prefetched_geometry_graph_matches_eager)?  Repeats the comparison and prints the largest |delta loss| per repetition."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.nn.functional as F

from pointnet12_amd import graph as G_
from pointnet12_amd import parallel
from pointnet12_amd import pointnet2 as M
from pointnet12_amd.graph import GraphedStep


def main(reps=8):
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(ROOT, "tests", "golden", "g6_nets.npz"))
    pts = torch.from_numpy(g["points"]).to(dev)
    labels = torch.from_numpy(g["labels"]).to(dev)
    for rep in range(reps):
        seqs = []
        for mode in ("eager", "eager", "prefetch", "prefetch-forked"):
            torch.manual_seed(int(g["init_seed"]))
            net = M.PointNet2SemSegMsg(13, 6)
            net.drop1.p = 0.0
            net.to(dev).train()
            bucket = parallel.FlatGradBucket(net)

            def compute():
                bucket.zero()
                lp = net(pts)
                loss = F.nll_loss(lp.reshape(-1, 13), labels.reshape(-1))
                G_.fork_point()
                loss.backward()
                return loss
            torch.manual_seed(31)
            if mode == "eager":
                for _ in range(2):
                    compute()
                step = compute
            else:
                step = GraphedStep(compute, dev, warmup=2, geometry_fn=lambda: net.features(pts), fork_in_step=mode == "prefetch-forked")
            seqs.append([float(step()) for _ in range(4)])
        a = np.array(seqs)
        print("rep %d: |eager - eager| %.2e   |eager - prefetch| %.2e   |eager - forked| %.2e   losses %s" %
              (rep, np.abs(a[0] - a[1]).max(), np.abs(a[0] - a[2]).max(), np.abs(a[0] - a[3]).max(), np.round(a[0], 6)))


if __name__ == "__main__":
    main()
