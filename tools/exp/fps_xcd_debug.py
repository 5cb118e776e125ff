"""Cooperative FPS with the partners of a cloud dealt onto one XCD (PN2_FPS_XCD_DEAL (an option of the experiment build: not in the shipped library)) against the plain launch order: same result, time."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from pointnet12_amd import _lib
import pointnet12_amd.pointnet_util as U
dev = torch.device("cuda:0")
torch.manual_seed(0)
for B, N, S in [(8, 65536, 1024), (16, 32768, 1024), (8, 65536, 8192)]:
    xyz = torch.rand(B, N, 3, device=dev)
    start = torch.zeros(B, dtype=torch.int64)
    ref = None
    for mode in (0, 1, 0, 1):
        _lib.set_option("PN2_FPS_XCD_DEAL (an option of the experiment build: not in the shipped library)", mode)
        U.farthest_point_sample(xyz, 64, start=start)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = U.farthest_point_sample(xyz, S, start=start)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if ref is None:
            ref = out.clone()
        print("B=%d N=%d npoint=%d  deal=%d  %.3f ms  equal: %s" % (B, N, S, mode, dt * 1e3, bool(torch.equal(out, ref))), flush=True)
