#!/usr/bin/env python3
"""FPS timing probe: python tools/bench_fps.py  (PN2_FPS_COOP=0 forces the single-workgroup kernel for N > 16384)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pointnet12_amd import pointnet_util as U
from pointnet12_amd import synthetic as syn

dev = torch.device("cuda:0")
for B, N, S in [(16, 4096, 1024), (8, 4096, 1024), (16, 1024, 256), (16, 2048, 512), (16, 8192, 1024), (8, 65536, 1024), (8, 65536, 8192), (1, 16384, 1024), (1, 20000, 1024), (1, 25000, 1024), (1, 28672, 1024), (8, 25000, 1024),
                (1, 65536, 1024)]:
    pts, _ = syn.kitti_batch(1, B, min(N, 65536))
    xyz = torch.from_numpy(pts[:, :3].transpose(0, 2, 1).copy()).to(dev)[:, :N].contiguous()
    start = torch.zeros(B, dtype=torch.int64, device=dev)
    for _ in range(2):
        U.farthest_point_sample(xyz, S, start)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    reps = 3
    for _ in range(reps):
        U.farthest_point_sample(xyz, S, start)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print("B=%3d N=%6d npoint=%5d  %9.3f ms  %6.2f us/iteration" % (B, N, S, ms, ms * 1e3 / S))
if "--uniform" in sys.argv:                 # no density contrast: what the spatial pruning gives when the cells are evenly filled
    for B, N, S in [(1, 16384, 1024), (1, 25000, 1024)]:
        xyz = torch.rand(B, N, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(0)) * torch.tensor([2.0, 2.0, 0.2], device=dev)
        start = torch.zeros(B, dtype=torch.int64, device=dev)
        for _ in range(2):
            U.farthest_point_sample(xyz, S, start)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            U.farthest_point_sample(xyz, S, start)
        b.record()
        torch.cuda.synchronize()
        print("uniform B=%3d N=%6d npoint=%5d  %6.2f us/iteration" % (B, N, S, a.elapsed_time(b) / 3 * 1e3 / S))
