#!/usr/bin/env python3
"""Round 6: the pooled last layer's forward with and without the store of its pre-BN output Y (pn2_conv1x1_fwd_pool, Y = NULL)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from pointnet12_amd import _lib
from pointnet12_amd._lib import ptr as p

SHAPES = [(1048576, 96, 128, 128), (524288, 64, 128, 64), (262144, 196, 256, 128), (131072, 128, 256, 64)]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev).manual_seed(0)
    for P, K, N, Kp in SHAPES:
        X = torch.randn(P, K, device=dev, generator=g)
        W = torch.randn(N, K, device=dev, generator=g)
        bias = torch.randn(N, device=dev, generator=g)
        gamma = torch.randn(N, device=dev, generator=g)
        Y = torch.empty(P, N, device=dev)
        aff = torch.zeros(4 * K, device=dev)
        aff[K:2 * K] = 1.0
        aff[3 * K:] = 1.0
        res = {}
        for name, y in (("store", Y), ("nostore", None)):
            stats = torch.zeros(8 * 2 * N, device=dev, dtype=torch.float64)
            ws = torch.zeros(2 * (P // Kp) * N, device=dev)

            def fn():
                rc = lib.pn2_conv1x1_fwd_pool(p(X), K, p(aff), p(W), K, p(bias), p(y), N, P, K, N, p(stats), Kp, p(gamma), p(ws), None, st)
                assert rc == 0, rc
            us = timeit(fn)
            stats.zero_()
            fn()
            torch.cuda.synchronize()
            res[name] = (us, stats.clone(), ws.clone())
        same = bool(torch.equal(res["store"][2], res["nostore"][2])) and bool(torch.allclose(res["store"][1], res["nostore"][1], rtol=1e-12))
        print("fwdpool (%d, %d -> %d, K=%d): store %.1f us, no store %.1f us; extrema / statistics equal: %s" %
              (P, K, N, Kp, res["store"][0], res["nostore"][0], same))


if __name__ == "__main__":
    main()
