#!/usr/bin/env python3
"""Micro-benchmark of pn2_group_affine_bwd (scatter of dY to the source points) under different index patterns."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointnet12_amd import _lib, pointnet_util as U, synthetic as syn
from pointnet12_amd._lib import ptr as p
lib = _lib.load(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
B, N0, N, S, C = 16, 4096, 512, 128, 128
pts = torch.from_numpy(syn.kitti_batch(0, B, N0)[0]).to(dev)
xyz0 = pts[:, :3].permute(0, 2, 1).contiguous()
fidx = U.farthest_point_sample(xyz0, N, torch.zeros(B, dtype=torch.long, device=dev))
xyz = U.index_points(xyz0, fidx, False)
new = U.index_points(xyz, U.farthest_point_sample(xyz, S, torch.zeros(B, dtype=torch.long, device=dev)), False)
for K, r in [(128, 0.8), (64, 0.4)]:
    P = B * S * K
    real = U.query_ball_point(r, K, xyz, new)
    rnd = torch.randint(0, N, (B, S, K), device=dev)
    perm = (torch.arange(K, device=dev).view(1, 1, K) + torch.arange(S, device=dev).view(1, S, 1) * 7) % N
    perm = perm.expand(B, S, K).contiguous()
    dZ, Y = torch.randn(P, C, device=dev), torch.randn(P, C, device=dev)
    coef = torch.ones(4 * C, device=dev)
    for name, idx in [("ball-query", real), ("random", rnd), ("strided", perm)]:
        uniq = np.mean([len(torch.unique(idx[0, s])) for s in range(0, S, 16)])
        G = torch.zeros(B * N, C, device=dev); dWx = torch.zeros(C, 3, device=dev)
        def fn():
            assert lib.pn2_group_affine_bwd(p(dZ), C, p(Y), C, p(coef), p(xyz), p(new), p(idx), B, N, S, K, C, p(G), C, p(dWx), 3, st) == 0
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize()
        t = a.elapsed_time(b) / 10
        print("K=%3d %-10s %7.1f us   %.0f GB/s of atomics, distinct idx per group %.1f" % (K, name, t * 1e3, P * C * 4 / t / 1e6, uniq))
