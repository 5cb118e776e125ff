#!/usr/bin/env python3
"""Diagnostic: where the waves of the LDS-DMA ring forward (csrc/mlp_wide.hip, ring_fwd_kernel) spend their cycles.
Needs a STAMP build of the library (make -C pointnet12_amd/csrc STAMP=1) loaded through PN2_LIB_PATH:
    PN2_LIB_PATH=pointnet12_amd/libpn2_hip_stamp.so python tools/stamp_wide.py
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
NAMES = (["barrier", "dma", "half1+xform", "wait", "epilogue", "half2"] if os.environ.get("PN2_RING") == "1" else
         ["barrier", "stage", "fetch", "mfma", "epilogue", "other"])         # ring_fwd_kernel / split_nt_kernel


def r4(c):
    return (c + 3) & ~3


def affine(c):
    a = torch.zeros(4 * r4(c), device=dev)
    a[:c] = rnd(c) * 0.1; a[r4(c):r4(c) + c] = 1.0 + rnd(c) * 0.1; a[2 * r4(c):2 * r4(c) + c] = rnd(c) * 0.1; a[3 * r4(c):3 * r4(c) + c] = 1.0
    return a


def dump():
    buf = (ctypes.c_ulonglong * (64 * 8 * 8))()
    raw.pn2_debug_stamps_wide(buf, 64 * 8 * 8)
    a = np.array(buf, dtype=np.float64).reshape(64, 8, 8)
    for w in range(8):
        v = a[:, w, :]
        v = v[v.sum(1) > 0]
        if not len(v):
            continue
        tot = v[:, :6].sum(1).mean()
        ghz = (v[:, 6] / np.maximum(v[:, 7], 1)).mean() * 0.1
        print("   wave %d  %8.0f cycles (%.0f us at the %.2f GHz it held): " % (w, tot, v[:, 7].mean() / 100.0, ghz)
              + "  ".join("%s %4.1f%%" % (n, 100 * x / tot) for n, x in zip(NAMES, v.mean(0)[:6])))


def dump_abs():
    buf = (ctypes.c_ulonglong * (256 * 8 * 4))()
    raw.pn2_debug_stamps_wide_abs(buf, 256 * 8 * 4)
    a = np.array(buf, dtype=np.float64).reshape(256, 8, 4) / 100.0          # us
    a = a[a[:, 0, 0] > 0]
    t0 = a[:, :, 0].min()
    a -= t0
    f = lambda x: "%.1f / %.1f / %.1f" % (x.min(), np.median(x), x.max())
    print("   absolute us after the first wave's entry (min / median / max over %d workgroups x waves): entry %s | loop start %s | loop end %s | exit %s"
          % (len(a), f(a[:, :, 0]), f(a[:, :, 1]), f(a[:, :, 2]), f(a[:, :, 3])))


for P, K, N in [(262144, 196, 256), (131072, 128, 256), (262144, 128, 196), (131072, 128, 128)]:
    X = torch.zeros(P, r4(K), device=dev); X[:, :K] = rnd(P, K)
    W, bias, Y = rnd(N, K), rnd(N), torch.empty(P, r4(N), device=dev)
    stats = torch.zeros(8 * 2 * N, device=dev, dtype=torch.float64)
    aff = affine(K)
    for _ in range(5):
        assert lib.pn2_conv1x1_fwd(p(X), r4(K), p(aff), p(W), K, p(bias), p(Y), r4(N), P, K, N, p(stats), None, None, st) == 0
    torch.cuda.synchronize()
    print("fwd", (P, K, N))
    dump()
    if os.environ.get("PN2_RING") != "1":
        dump_abs()

# the data gradients on the same kernel (split_nt_kernel, EPI_MASK): dense dZ, or the max-pool's sparse dZp / arg (groups of Kp rows)
if os.environ.get("PN2_RING") != "1":
    for P, Cl, Cp, Kp in [(262144, 256, 196, 128), (131072, 256, 128, 64), (262144, 196, 128, 0), (131072, 128, 128, 0)]:
        Y = torch.zeros(P, r4(Cl), device=dev); Y[:, :Cl] = rnd(P, Cl)
        Yp = torch.zeros(P, r4(Cp), device=dev); Yp[:, :Cp] = rnd(P, Cp)
        coef, affp = affine(Cl), affine(Cp)
        Wt = rnd(Cl, Cp)
        if Kp:
            G = P // Kp
            dOut = torch.zeros(G, r4(Cl), device=dev); dOut[:, :Cl] = rnd(G, Cl)
            arg = torch.randint(0, Kp, (G, r4(Cl)), device=dev, dtype=torch.int32, generator=g)
            dz = (None, 0, p(dOut), r4(Cl), p(arg), Kp)
        else:
            dZ = torch.zeros(P, r4(Cl), device=dev); dZ[:, :Cl] = rnd(P, Cl)
            dz = (p(dZ), r4(Cl), None, 0, None, 0)
        dX = torch.empty(P, r4(Cp), device=dev)
        red = torch.zeros(8 * 2 * Cp, device=dev, dtype=torch.float64)
        for _ in range(5):
            assert lib.pn2_conv1x1_dgrad(*dz, p(Y), r4(Cl), p(coef), p(Wt), Cp, p(Yp), r4(Cp), p(affp), p(dX), r4(Cp), p(red), P, Cl, Cp,
                                         None, None, st) == 0
        torch.cuda.synchronize()
        print("dgrad", (P, Cl, Cp, Kp))
        dump()
        dump_abs()
        del Y, Yp, dX

# the full-tile weight gradients (split_tn_kernel)
if os.environ.get("PN2_RING") != "1":
    for P, Cl, Cp, Kp in [(262144, 256, 196, 128), (131072, 256, 128, 64), (262144, 196, 128, 0)]:
        Y = torch.zeros(P, r4(Cl), device=dev); Y[:, :Cl] = rnd(P, Cl)
        X = torch.zeros(P, r4(Cp), device=dev); X[:, :Cp] = rnd(P, Cp)
        coef, affx = affine(Cl), affine(Cp)
        if Kp:
            G = P // Kp
            dOut = torch.zeros(G, r4(Cl), device=dev); dOut[:, :Cl] = rnd(G, Cl)
            arg = torch.randint(0, Kp, (G, r4(Cl)), device=dev, dtype=torch.int32, generator=g)
            dz = (None, 0, p(dOut), r4(Cl), p(arg), Kp)
        else:
            dZ = torch.zeros(P, r4(Cl), device=dev); dZ[:, :Cl] = rnd(P, Cl)
            dz = (p(dZ), r4(Cl), None, 0, None, 0)
        dW = torch.zeros(Cl, Cp, device=dev)
        for _ in range(5):
            assert lib.pn2_conv1x1_wgrad(*dz, p(Y), r4(Cl), p(coef), p(X), r4(Cp), p(affx), p(dW), Cp, None, P, Cl, Cp, None, st) == 0
        torch.cuda.synchronize()
        print("wgrad", (P, Cl, Cp, Kp))
        dump()
        dump_abs()
        del Y, X
