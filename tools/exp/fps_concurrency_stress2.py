#!/usr/bin/env python3
"""Round 6: WHICH co-running kernel breaks pn2_fps?  FPS (16 x 4096 -> 512) on a side stream beside ONE kind of main-stream kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from pointnet12_amd import _lib
from pointnet12_amd import pointnet_util as U
from pointnet12_amd import synthetic as syn
from pointnet12_amd._lib import ptr as p


def main(trials=120):
    dev = torch.device("cuda:0")
    lib = _lib.load()
    big_np, _ = syn.kitti_batch(0, 16, 4096)
    xyz = torch.from_numpy(np.ascontiguousarray(big_np[:, :3, :].transpose(0, 2, 1))).to(dev)
    B, N, npoint = 16, 4096, 512
    P = 1 << 19
    g = torch.Generator(device=dev).manual_seed(0)
    X64, X96 = torch.randn(P, 64, device=dev, generator=g), torch.randn(P, 96, device=dev, generator=g)
    W96, W128 = torch.randn(96, 64, device=dev, generator=g), torch.randn(128, 96, device=dev, generator=g)
    b96, b128 = torch.randn(96, device=dev, generator=g), torch.randn(128, device=dev, generator=g)
    Y96, Y128 = torch.empty(P, 96, device=dev), torch.empty(P, 128, device=dev)
    aff64 = torch.zeros(4 * 64, device=dev); aff64[64:128] = 1; aff64[192:] = 1
    aff96 = torch.zeros(4 * 96, device=dev); aff96[96:192] = 1; aff96[288:] = 1
    st96 = torch.zeros(8 * 2 * 96, device=dev, dtype=torch.float64)
    st128 = torch.zeros(8 * 2 * 128, device=dev, dtype=torch.float64)
    ws = torch.zeros(2 * (P // 128) * 128, device=dev)
    big = torch.randn(1 << 26, device=dev)
    main_s = torch.cuda.current_stream().cuda_stream

    def k_elementwise():
        big.mul_(1.0001)

    def k_fwd_plain():          # gemm / split forward 64 -> 96, no input BatchNorm
        assert lib.pn2_conv1x1_fwd(p(X64), 64, None, p(W96), 64, p(b96), p(Y96), 96, P, 64, 96, p(st96), None, None, main_s) == 0

    def k_fwd_bn():             # split_nt forward 64 -> 96 with BN + ReLU in the loader
        assert lib.pn2_conv1x1_fwd(p(X64), 64, p(aff64), p(W96), 64, p(b96), p(Y96), 96, P, 64, 96, p(st96), None, None, main_s) == 0

    def k_fwd_pool():           # split_nt pooled forward 96 -> 128 (two four-wave workgroups per CU)
        assert lib.pn2_conv1x1_fwd_pool(p(X96), 96, p(aff96), p(W128), 96, p(b128), p(Y128), 128, P, 96, 128, p(st128), 128, p(b128), p(ws), None, main_s) == 0

    def k_fwd_pool_nostore():
        assert lib.pn2_conv1x1_fwd_pool(p(X96), 96, p(aff96), p(W128), 96, p(b128), None, 128, P, 96, 128, p(st128), 128, p(b128), p(ws), None, main_s) == 0

    def k_fwd_pool_wg1():
        _lib.set_option("PN2_SPLIT_WG2", 0)
        k_fwd_pool()
        _lib.set_option("PN2_SPLIT_WG2", 1)

    def k_fwd_pool_res():       # the fp32 weight-resident pooled forward instead (PN2_SPLIT_NARROW = 0)
        _lib.set_option("PN2_SPLIT_NARROW", 0)
        k_fwd_pool()
        _lib.set_option("PN2_SPLIT_NARROW", 1)

    def k_fwd_96_128_plainepi():  # the same GEMM without the pooling epilogue
        assert lib.pn2_conv1x1_fwd(p(X96), 96, p(aff96), p(W128), 96, p(b128), p(Y128), 128, P, 96, 128, p(st128), None, None, main_s) == 0

    A_ = torch.randn(8192, 8192, device=dev)
    B_ = torch.randn(8192, 8192, device=dev)
    sm = torch.randn(1 << 16, 1024, device=dev)

    def k_torch_sum():
        for _ in range(8):
            big.sum()

    def k_torch_softmax():
        for _ in range(4):
            torch.softmax(sm, 1)

    def k_torch_matmul():
        torch.mm(A_, B_)

    def k_ball():
        new = xyz[:, :1024].contiguous()
        U.query_ball_point(0.2, 64, xyz, new)
        U.three_nn(xyz, new)

    side = torch.cuda.Stream(device=dev)
    for name, kern in (("torch sum (LDS reduction)", k_torch_sum), ("torch softmax", k_torch_softmax), ("torch mm (rocBLAS)", k_torch_matmul),
                       ("ball query + 3-NN (this library)", k_ball), ("elementwise (no LDS)", k_elementwise), ("fwd 64->96 BN input (split_nt)", k_fwd_bn),
                       ("fwd 96->128 pooled (split_nt, 2 WG/CU)", k_fwd_pool), ("fwd 96->128 pooled, no store", k_fwd_pool_nostore),
                       ("fwd 96->128 pooled, 1 WG/CU", k_fwd_pool_wg1), ("fwd 96->128 pooled, fp32 resident kernel", k_fwd_pool_res),
                       ("fwd 96->128 without pooling", k_fwd_96_128_plainepi)):
        gen = torch.Generator().manual_seed(1)
        bad = 0
        for tr in range(trials):
            start = torch.randint(0, N, (B,), generator=gen).to(dev)
            ref = U.farthest_point_sample(xyz, npoint, start).clone()
            torch.cuda.synchronize()
            side.wait_stream(torch.cuda.current_stream())
            kern()
            with torch.cuda.stream(side):
                conc = U.farthest_point_sample(xyz, npoint, start).clone()
            kern(); kern()
            torch.cuda.synchronize()
            bad += int(not torch.equal(ref, conc))
        extra = ""
        if os.environ.get("PN2_LIB_PATH"):
            import ctypes
            raw = ctypes.CDLL(os.environ["PN2_LIB_PATH"])
            if hasattr(raw, "pn2_fps_check_read"):
                buf = (ctypes.c_uint * 4)()
                raw.pn2_fps_check_read(buf)
                extra = "   [LDS mirror check: %d corrupted points in %d workgroup runs so far; re-reads of cloud[far] that differed: %d]" % (buf[0], buf[1], buf[2])
        print("beside %-45s: %d of %d FPS results differ%s" % (name, bad, trials, extra))


if __name__ == "__main__":
    main()
