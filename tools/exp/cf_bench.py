#!/usr/bin/env python3
"""First-layer weight gradient: general skinny kernel (reads dZ, Y, X) vs row moments + closed form (dZ, X)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pointnet12_amd import _lib
from pointnet12_amd._lib import ptr as p
lib = _lib.load(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for P, M, N in [(1048576, 64, 9), (524288, 64, 9), (262144, 32, 9), (524288, 32, 9)]:
    dZ, Y = rnd(P, M), rnd(P, M)
    X = torch.zeros(P, 12, device=dev); X[:, :N] = rnd(P, N)
    coef = torch.zeros(4 * M, device=dev); coef[:M] = 1.0; coef[M:2 * M] = 0.1
    W, b = rnd(M, N), rnd(M)
    dW = torch.zeros(M, N, device=dev)
    mom = torch.zeros(int(lib.pn2_conv1x1_wgrad_cf_scratch_bytes()), device=dev, dtype=torch.uint8)
    t0 = timeit(lambda: lib.pn2_conv1x1_wgrad(p(dZ), M, None, 0, None, 0, p(Y), M, p(coef), p(X), 12, None, p(dW), N, None, P, M, N, None, st))
    t1 = 0.0
    t2 = timeit(lambda: lib.pn2_conv1x1_wgrad_cf(p(dZ), M, p(coef), p(X), 12, p(W), N, p(b), p(mom), p(dW), N, P, M, N, None, st))
    print("P=%8d M=%3d N=%2d  general %6.1f us   moments %6.1f us   closed form %6.1f us" % (P, M, N, t0, t1, t2))
