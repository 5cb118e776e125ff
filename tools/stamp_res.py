#!/usr/bin/env python3
"""Diagnostic: where the waves of the weight-resident kernels spend their cycles.  Needs a STAMP build:
    make -C pointnet12_amd/csrc clean; make -C pointnet12_amd/csrc STAMP=1 -j4     (rebuild without STAMP afterwards)
    python tools/stamp_res.py
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointnet12_amd import _lib
from pointnet12_amd._lib import ptr as p
lib = _lib.load(); raw = ctypes.CDLL(_lib.LIB_PATH)
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)


def affine(c):
    a = torch.zeros(4 * c, device=dev)
    a[:c] = rnd(c) * 0.1; a[c:2 * c] = 1.0 + rnd(c) * 0.1; a[2 * c:3 * c] = rnd(c) * 0.1; a[3 * c:] = 1.0
    return a


def dump(names):
    buf = (ctypes.c_ulonglong * (64 * 8 * 8))()
    raw.pn2_debug_stamps_res(buf, 64 * 8 * 8)
    a = np.array(buf, dtype=np.float64).reshape(64, 8, 8)
    for w in range(8):
        v = a[:, w, :]
        v = v[v.sum(1) > 0]
        if not len(v):
            continue
        tot = v.sum(1).mean()
        print("   wave %d  %8.0f cycles: " % (w, tot) + "  ".join("%s %4.1f%%" % (n, 100 * x / tot) for n, x in zip(names, v.mean(0)) if x > 0))


for P, Cl, Cp, Kp in [(1048576, 128, 96, 128), (1048576, 96, 64, 0), (524288, 128, 64, 64), (524288, 64, 64, 0)]:
    Y, Yp = rnd(P, Cl), rnd(P, Cp)
    coef, affp = affine(Cl), affine(Cp)
    Wt = rnd(Cl, Cp)
    if Kp:
        G = P // Kp
        dOut = rnd(G, Cl); arg = torch.randint(0, Kp, (G, Cl), device=dev, dtype=torch.int32, generator=g)
        dz = (None, 0, p(dOut), Cl, p(arg), Kp)
    else:
        dZ = rnd(P, Cl); dz = (p(dZ), Cl, None, 0, None, 0)
    dX = torch.empty(P, Cp, device=dev); red = torch.zeros(16 * Cp, device=dev, dtype=torch.float64); dW = torch.zeros(Cl, Cp, device=dev)
    for _ in range(3):
        assert lib.pn2_conv1x1_bwd(*dz, p(Y), Cl, p(coef), p(Wt), Cp, p(Yp), Cp, p(affp), p(dX), Cp, p(red), p(dW), Cp, P, Cl, Cp, None, st) == 0
    print("bwd", (P, Cl, Cp, Kp))
    dump(["finish", "fetch", "barrier1", "compute", "barrier2", "", "", ""])
    del Y, Yp, dX
for P, K, N in [(1048576, 96, 128), (1048576, 64, 96)]:
    X = rnd(P, K); W = rnd(N, K); b = rnd(N); Y = torch.empty(P, N, device=dev)
    stats = torch.zeros(16 * N, device=dev, dtype=torch.float64); aff = affine(K)
    for _ in range(3):
        assert lib.pn2_conv1x1_fwd(p(X), K, p(aff), p(W), K, p(b), p(Y), N, P, K, N, p(stats), None, None, st) == 0
    print("fwd", (P, K, N))
    dump(["transform", "fetch", "mfma", "epilogue(last)", "epilogue", "", "", ""])
