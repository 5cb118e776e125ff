#!/usr/bin/env python3
"""Round 6: pn2_fps (16 x 4096 -> 512) on a side stream while whole eager MSG-SemSeg training steps run on the main stream."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from pointnet12_amd import _lib, parallel
from pointnet12_amd import pointnet2 as M
from pointnet12_amd import pointnet_util as U
from pointnet12_amd import synthetic as syn
from pointnet12_amd.loss import nll_loss


def main(trials=40):
    dev = torch.device("cuda:0")
    pts_np, lab_np = syn.kitti_batch(0, 16, 4096)
    pts, labels = torch.from_numpy(pts_np).to(dev), torch.from_numpy(lab_np).to(dev)
    xyz = torch.from_numpy(np.ascontiguousarray(pts_np[:, :3, :].transpose(0, 2, 1))).to(dev)
    torch.manual_seed(0)
    net = M.PointNet2SemSegMsg(13, 6).to(dev).train()
    bucket = parallel.FlatGradBucket(net, direct=True)

    def step():
        bucket.zero()
        lp = net(pts)
        nll_loss(lp.reshape(-1, 13), labels.reshape(-1)).backward()
    for _ in range(2):
        step()
    side = torch.cuda.Stream(device=dev)
    gen = torch.Generator().manual_seed(1)
    bad = 0
    for tr in range(trials):
        start = torch.randint(0, 4096, (16,), generator=gen).to(dev)
        ref = U.farthest_point_sample(xyz, 512, start).clone()
        torch.cuda.synchronize()
        side.wait_stream(torch.cuda.current_stream())
        outs = []
        with torch.cuda.stream(side):
            for _ in range(12):                       # FPS launches back to back under the whole step
                outs.append(U.farthest_point_sample(xyz, 512, start).clone())
        step()
        torch.cuda.synchronize()
        bad += sum(int(not torch.equal(ref, o)) for o in outs)
    print("options %s: %d of %d FPS results beside eager MSG steps differ" % ({k: v for k, v in _lib.options().items() if os.environ.get(k)}, bad, trials * 12))


if __name__ == "__main__":
    main()
