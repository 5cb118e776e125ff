#!/usr/bin/env python3
"""Ball query at the level sizes of BASELINE cfg5 (B=8 x 65 536 points, MSG x16: 16 384 and 4096 centres), KITTI-shaped synthetic
clouds, centres = FPS samples.  PN2_BQ_ORDER=0 in the environment: centres in sampling order (A/B)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnet12_amd import pointnet_util as U      # noqa: E402
from pointnet12_amd import synthetic as syn      # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B = 8
    pts, _ = syn.kitti_batch(5, B, 65536)
    xyz = torch.from_numpy(np.ascontiguousarray(pts[:, :3].transpose(0, 2, 1))).to(dev)
    start = torch.zeros(B, dtype=torch.int64, device=dev)
    lvl = [(xyz, 16384, [(0.05, 16), (0.1, 32)])]
    f1 = U.farthest_point_sample(xyz, 16384, start)
    l1 = U.index_points(xyz, f1)
    lvl.append((l1, 4096, [(0.1, 16), (0.2, 32)]))
    total = 0.0
    for src, S, cases in lvl:
        new = U.index_points(src, U.farthest_point_sample(src, S, start))
        for r, k in cases:
            for _ in range(2):
                U.query_ball_point(r, k, src, new)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                U.query_ball_point(r, k, src, new)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 100
            total += ms
            print("N=%6d S=%5d r=%.2f K=%3d   %.3f ms" % (src.shape[1], S, r, k, ms))
    print("sum %.3f ms  (PN2_BQ_ORDER=%s)" % (total, os.environ.get("PN2_BQ_ORDER", "1")))


if __name__ == "__main__":
    main()
