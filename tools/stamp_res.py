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
        tot = v[:, :6].sum(1).mean()
        ghz = (v[:, 6] / np.maximum(v[:, 7], 1)).mean() * 0.1          # shader cycles per 100 MHz tick
        print("   wave %d  %8.0f cycles (%.0f us at the %.2f GHz it held): " % (w, tot, v[:, 7].mean() / 100.0, ghz)
              + "  ".join("%s %4.1f%%" % (n, 100 * x / tot) for n, x in zip(names, v.mean(0)[:6]) if x > 0))


def dump_abs():
    buf = (ctypes.c_ulonglong * (64 * 8 * 4))()
    raw.pn2_debug_stamps_abs(buf, 64 * 8 * 4)
    a = np.array(buf, dtype=np.float64).reshape(64, 8, 4) / 100.0          # us
    t0 = a[:, :, 0].min()
    a -= t0
    print("   absolute (us after the first wave's entry; min / median / max over 64 workgroups x 8 waves): entry %.1f/%.1f/%.1f  loop start %.1f/%.1f/%.1f"
          "  loop end %.1f/%.1f/%.1f  exit %.1f/%.1f/%.1f" % tuple(f(a[:, :, i]) for i in range(4) for f in (np.min, np.median, np.max)))
    print("   loop end by wave (median over workgroups): " + " ".join("%.1f" % np.median(a[:, w, 2]) for w in range(8)))
    print("   workgroup exit (max over its waves), sorted: " + " ".join("%.0f" % x for x in np.sort(a[:, :, 3].max(1))))


for P, Cl, Cp, Kp in [(1048576, 128, 96, 128), (1048576, 96, 64, 0), (524288, 128, 64, 64), (524288, 64, 64, 0), (524288, 32, 32, 0)]:
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
    dump(["stage/finish", "fetch", "barrier", "mfma/compute", "epilogue/barrier2", "", "", ""])
    dump_abs()
    del Y, Yp, dX
for P, K, N in [(1048576, 96, 128), (1048576, 64, 96)]:
    X = rnd(P, K); W = rnd(N, K); b = rnd(N); Y = torch.empty(P, N, device=dev)
    stats = torch.zeros(16 * N, device=dev, dtype=torch.float64); aff = affine(K)
    for _ in range(3):
        assert lib.pn2_conv1x1_fwd(p(X), K, p(aff), p(W), K, p(b), p(Y), N, P, K, N, p(stats), None, None, st) == 0
    import time
    torch.cuda.synchronize(); t_0 = time.perf_counter()
    for _ in range(20):
        lib.pn2_conv1x1_fwd(p(X), K, p(aff), p(W), K, p(b), p(Y), N, P, K, N, p(stats), None, None, st)
    torch.cuda.synchronize()
    print("fwd", (P, K, N), "%.1f us per launch, 20 back to back" % ((time.perf_counter() - t_0) / 20 * 1e6))
    dump(["transform", "fetch", "mfma", "epilogue(last)", "epilogue", "", "", ""])
    dump_abs()
