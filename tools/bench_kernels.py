#!/usr/bin/env python3
"""Micro-benchmark of the shared-MLP GEMM entry points on the layer shapes of MSG-SemSeg (B=16 x 4096).

    python tools/bench_kernels.py [fwd|dgrad|wgrad|all] [--reps 20]

Prints per shape: time, TFLOP/s (2*P*K*N), algorithmic GB/s, and the larger of the two roofline fractions.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from pointnet12_amd import _lib
from pointnet12_amd._lib import ptr as p

FWD = [(1048576, 9, 64), (1048576, 64, 96), (1048576, 96, 128), (524288, 64, 64), (524288, 64, 128), (262144, 323, 128),
       (262144, 128, 196), (262144, 196, 256), (131072, 323, 128), (131072, 128, 256), (262144, 32, 64), (65536, 128, 128), (131072, 128, 128),
       (65536, 137, 128), (8192, 576, 256), (8192, 256, 128), (8192, 320, 128), (2048, 515, 256), (2048, 256, 512), (2048, 512, 1024), (2048, 1536, 256),
       (2048, 256, 256)]
# backward shapes: (P, C_l, C_{l-1}, pooled K or 0)
BWD = [(16384, 256, 320, 0), (16384, 128, 256, 0), (4096, 256, 384, 0), (4096, 256, 256, 0), (1024, 256, 768, 0), (8192, 512, 256, 32), (8192, 256, 256, 0),
       (1048576, 64, 9, 0), (524288, 64, 9, 0), (262144, 32, 9, 0), (1048576, 128, 96, 128), (1048576, 96, 64, 0), (262144, 256, 196, 128), (262144, 196, 128, 0), (262144, 128, 323, 0),
       (524288, 128, 64, 64), (524288, 64, 64, 0), (131072, 256, 128, 64), (65536, 128, 128, 0),
       (131072, 128, 128, 0), (65536, 128, 137, 0), (262144, 64, 32, 32), (262144, 32, 32, 0), (524288, 64, 32, 32), (524288, 32, 32, 0), (32768, 256, 256, 0), (32768, 256, 320, 0),
       (8192, 128, 256, 0), (8192, 256, 576, 0), (8192, 128, 320, 0), (2048, 256, 256, 0), (2048, 256, 1536, 0), (2048, 1024, 512, 128), (2048, 512, 256, 0),
       (2048, 256, 515, 0)]


def r4(c):
    return (c + 3) & ~3


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def report(tag, shape, secs, flops, nbytes):
    """One line per shape, priced as bench.py prices a kernel: algorithmic bytes against HBM 8 TB/s, algorithmic fp32 flops against
    the matrix peak of the pipe the kernel that just ran uses (pn2_last_kernel(): f32 157.3 TF, bf16x3 split 416.7 TF); the larger
    fraction is the bound -- never above 1."""
    import bench
    kern = _lib.load().pn2_last_kernel()
    key = bench.kernel_key(kern.decode()) if kern else ""
    pipe = bench.kernel_pipe(key) if key else "f32"
    bound, ach, peak, unit, frac, hf, mf = bench.price(flops, nbytes, secs, pipe)
    assert frac <= 1.0, (tag, shape, frac)
    tf, gbs = flops / secs / 1e12, nbytes / secs / 1e9
    print("%-6s %-28s %8.1f us %7.2f TF %8.1f GB/s  %-4s frac %.3f  (hbm %.3f, %s mfma %.3f)  %s" %
          (tag, shape, secs * 1e6, tf, gbs, bound, frac, hf, pipe, mf, key[:70]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="?", default="all")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="", help="comma list of row counts P to keep (profiling runs)")
    ap.add_argument("--shape", default="", help="keep only this shape: P,K,N (forward) / P,C_l,C_prev (backward)")
    args = ap.parse_args()
    if args.only:
        keep = {int(x) for x in args.only.split(",")}
        FWD[:] = [s_ for s_ in FWD if s_[0] in keep]
        BWD[:] = [s_ for s_ in BWD if s_[0] in keep]
    if args.shape:
        want = tuple(int(x) for x in args.shape.split(","))
        FWD[:] = [s_ for s_ in FWD if s_[:3] == want]
        BWD[:] = [s_ for s_ in BWD if s_[:3] == want]
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev).manual_seed(0)

    def rnd(*s):
        return torch.randn(*s, device=dev, generator=g)

    def affine(c):
        a = torch.zeros(4 * r4(c), device=dev)
        a[:c] = rnd(c) * 0.1
        a[r4(c):r4(c) + c] = 1.0 + rnd(c) * 0.1
        a[2 * r4(c):2 * r4(c) + c] = rnd(c) * 0.1
        a[3 * r4(c):3 * r4(c) + c] = 1.0
        return a

    if args.which in ("fwd", "all"):
        for P, K, N in FWD:
            X = torch.zeros(P, r4(K), device=dev)
            X[:, :K] = rnd(P, K)
            W = rnd(N, K)
            bias, Y = rnd(N), torch.empty(P, r4(N), device=dev)
            stats = torch.zeros(8 * 2 * N, device=dev, dtype=torch.float64)
            aff = affine(K) if K > 12 else None

            def fn():
                rc = lib.pn2_conv1x1_fwd(p(X), r4(K), p(aff), p(W), K, p(bias), p(Y), r4(N), P, K, N, p(stats), None, None, st)
                assert rc == 0
            report("fwd", (P, K, N), timeit(fn, args.reps), 2.0 * P * K * N, 4.0 * (P * K + P * N + N * K))
            for Kp in (16, 32, 64):              # the same GEMM with the pooling extrema recorded in its epilogue
                if aff is None or not lib.pn2_res_supported(P, N, K) or N % 32 or K % 32:
                    continue
                ws = torch.empty(4 * (P // Kp) * N, device=dev)
                if lib.pn2_conv1x1_fwd_pool(p(X), r4(K), p(aff), p(W), K, p(bias), p(Y), r4(N), P, K, N, p(stats), Kp, p(bias), p(ws), None, st) != 0:
                    continue                     # W plus eight staging buffers exceed LDS

                def fnp():
                    rc = lib.pn2_conv1x1_fwd_pool(p(X), r4(K), p(aff), p(W), K, p(bias), p(Y), r4(N), P, K, N, p(stats), Kp, p(bias), p(ws), None, st)
                    assert rc == 0
                report("fwdpool%d" % Kp, (P, K, N), timeit(fnp, args.reps), 2.0 * P * K * N, 4.0 * (P * K + P * N + N * K))
            del X, Y

    if args.which in ("bwdcf", "all"):
        for P, Cl, Cp, Kp in BWD:
            if not Kp or not lib.pn2_conv1x1_bwd_cf_supported(P, Cl, Cp, Kp):
                continue
            Yp, coef, affp, Wt, bias = rnd(P, r4(Cp)), affine(Cl), affine(Cp), rnd(Cl, Cp), rnd(Cl)
            G = P // Kp
            dOut = rnd(G, r4(Cl))
            arg = torch.randint(0, Kp, (G, r4(Cl)), device=dev, dtype=torch.int32, generator=g)
            dX = torch.empty(P, r4(Cp), device=dev)
            red = torch.zeros(8 * 2 * Cp, device=dev, dtype=torch.float64)
            dW = torch.zeros(Cl, Cp, device=dev)
            scratch = torch.empty(int(lib.pn2_conv1x1_bwd_cf_scratch_bytes(Cl, Cp)), device=dev, dtype=torch.uint8)

            def fn():
                rc = lib.pn2_conv1x1_bwd_cf(p(dOut), r4(Cl), p(arg), Kp, p(coef), p(Wt), Cp, p(bias), p(Yp), r4(Cp), p(affp), p(dX), r4(Cp), p(red),
                                            p(dW), Cp, P, Cl, Cp, None, p(scratch), st)
                assert rc == 0
            report("bwdcf", (P, Cl, Cp, Kp), timeit(fn, args.reps), 4.0 * P * Cl * Cp, 4.0 * (2 * P * Cp + 2 * Cl * Cp))
            del Yp, dX

    for which in ("dgrad", "wgrad", "bwd", "pair"):
        if args.which not in (which, "all"):
            continue
        for P, Cl, Cp, Kp in BWD:
            if which == "bwd" and not lib.pn2_res_supported(P, Cl, Cp):
                continue
            if which == "pair" and (P > 131072 or Cp < 64):
                continue
            Y, Yp = rnd(P, r4(Cl)), rnd(P, r4(Cp))
            coef, affp = affine(Cl), affine(Cp)
            Wt = rnd(Cl, Cp)                    # the Conv weight [C_l, C_{l-1}] as stored; dgrad reads it down the columns
            if Kp:
                G = P // Kp
                dOut, out = rnd(G, r4(Cl)), rnd(G, r4(Cl))
                arg = torch.randint(0, Kp, (G, r4(Cl)), device=dev, dtype=torch.int32, generator=g)
                dz = (None, 0, p(dOut), r4(Cl), p(arg), Kp)
                dy_bytes = P * Cl
            else:
                dZ = rnd(P, r4(Cl))
                dz = (p(dZ), r4(Cl), None, 0, None, 0)
                dy_bytes = 2 * P * Cl
            if which == "dgrad":
                dX = torch.empty(P, r4(Cp), device=dev)
                red = torch.zeros(8 * 2 * Cp, device=dev, dtype=torch.float64)

                def fn():
                    rc = lib.pn2_conv1x1_dgrad(*dz, p(Y), r4(Cl), p(coef), p(Wt), Cp, p(Yp), r4(Cp), p(affp), p(dX), r4(Cp),
                                               p(red), P, Cl, Cp, None, None, st)
                    assert rc == 0
                report("dgrad", (P, Cl, Cp, Kp), timeit(fn, args.reps), 2.0 * P * Cl * Cp, 4.0 * (dy_bytes + 2 * P * Cp))
            elif which == "pair":              # dgrad + wgrad as one call (one launch where the pair kernel takes the shape)
                dX = torch.empty(P, r4(Cp), device=dev)
                red = torch.zeros(8 * 2 * Cp, device=dev, dtype=torch.float64)
                dW = torch.zeros(Cl, Cp, device=dev)

                def fn():
                    rc = lib.pn2_conv1x1_bwd_pair(*dz, p(Y), r4(Cl), p(coef), p(Wt), Cp, p(Yp), r4(Cp), p(affp), p(dX), r4(Cp), p(red),
                                                  p(Yp), r4(Cp), p(affp), p(dW), Cp, P, Cl, Cp, None, st)
                    assert rc in (0, 1)                 # (1: run as dgrad + wgrad launches -- the kernel named is then the second one)
                report("pair", (P, Cl, Cp, Kp), timeit(fn, args.reps), 4.0 * P * Cl * Cp, 4.0 * (dy_bytes + 2 * P * Cp))
            elif which == "bwd":
                dX = torch.empty(P, r4(Cp), device=dev)
                red = torch.zeros(8 * 2 * Cp, device=dev, dtype=torch.float64)
                dW = torch.zeros(Cl, Cp, device=dev)

                def fn():
                    rc = lib.pn2_conv1x1_bwd(*dz, p(Y), r4(Cl), p(coef), p(Wt), Cp, p(Yp), r4(Cp), p(affp), p(dX), r4(Cp), p(red),
                                             p(dW), Cp, P, Cl, Cp, None, st)
                    assert rc == 0
                report("bwd", (P, Cl, Cp, Kp), timeit(fn, args.reps), 4.0 * P * Cl * Cp, 4.0 * (dy_bytes + 2 * P * Cp))
            else:
                dW = torch.zeros(Cl, Cp, device=dev)
                wsb = lib.pn2_conv1x1_wgrad_workspace_bytes(P, Cl, Cp, 1 if Kp else 0)      # > 0: two-phase dW flush (PN2_WGRAD_TWO_PHASE=1)
                ws = torch.empty(wsb, device=dev, dtype=torch.uint8) if wsb else None

                def fn():
                    if ws is not None:
                        rc = lib.pn2_conv1x1_wgrad_ws(*dz, p(Y), r4(Cl), p(coef), p(Yp), r4(Cp), p(affp) if Cp >= 16 else None, p(dW), Cp,
                                                      None, P, Cl, Cp, None, p(ws), st)
                    else:
                        rc = lib.pn2_conv1x1_wgrad(*dz, p(Y), r4(Cl), p(coef), p(Yp), r4(Cp), p(affp) if Cp >= 16 else None, p(dW), Cp,
                                                   None, P, Cl, Cp, None, st)
                    assert rc == 0
                report("wgrad", (P, Cl, Cp, Kp), timeit(fn, args.reps), 2.0 * P * Cl * Cp, 4.0 * (dy_bytes + P * Cp))


if __name__ == "__main__":
    main()
