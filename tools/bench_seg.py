#!/usr/bin/env python3
"""Micro-benchmark of pn2_group_affine_bwd_seg / pn2_three_interp_bwd_seg on the MSG-SemSeg shapes (real ball-query
and 3-NN indices of KITTI-shaped clouds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointnet12_amd import _lib, pointnet_util as U, synthetic as syn
from pointnet12_amd._lib import ptr as p

lib = _lib.load(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


B, N0 = 16, 4096
pts = torch.from_numpy(syn.kitti_batch(0, B, N0)[0]).to(dev)
xyz0 = pts[:, :3].permute(0, 2, 1).contiguous()
z = torch.zeros(B, dtype=torch.long, device=dev)
for (N, S, C, cases) in [(512, 128, 128, [(128, 0.8), (64, 0.4)]), (4096, 512, 64, [(128, 0.4), (64, 0.2)]), (4096, 1024, 32, [(32, 0.1)])]:
    xyz = xyz0 if N == N0 else U.index_points(xyz0, U.farthest_point_sample(xyz0, N, z), False)
    new = U.index_points(xyz, U.farthest_point_sample(xyz, S, z), False)
    for K, r in cases:
        P = B * S * K
        idx = U.query_ball_point(r, K, xyz, new)
        members, owners = U._inverse_index(idx.reshape(B, S * K), N)
        dZ, Y = torch.randn(P, C, device=dev), torch.randn(P, C, device=dev)
        coef = torch.ones(4 * C, device=dev)
        G = torch.zeros(B * N, C, device=dev); dWx = torch.zeros(C, 3, device=dev)
        scratch = torch.zeros(32 * 3 * C, device=dev)
        for tag, sc in (("direct", None), ("replicas", scratch)):
            def fn():
                assert lib.pn2_group_affine_bwd_seg(p(dZ), C, p(Y), C, p(coef), p(xyz), p(new), p(members), p(owners), B, N, S, K, C,
                                                    p(G), C, p(dWx), 3, p(sc), st) == 0
            t = timed(fn)
            print("group_affine_bwd_seg N=%4d S=%4d K=%3d C=%3d %-8s %7.1f us  %6.0f GB/s (dZ + Y rows)" % (N, S, K, C, tag, t, 2 * P * C * 4 / t / 1e3))
