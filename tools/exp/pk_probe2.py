#!/usr/bin/env python3
"""Round 6: tools/exp/pk_probe2.hip (variants of the reproducer) beside the pooled bf16-split forward.  python tools/exp/pk_probe2.py [var ...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from pointnet12_amd import _lib
from pointnet12_amd import synthetic as syn
from pointnet12_amd._lib import ptr as p

BITS = {1: "A scalar", 2: "B packed", 4: "branch-free", 8: "read before use", 16: "no minima", 32: "B unfenced", 64: "32-bit read of x, y, z first", 128: "32-bit read of x first", 256: "A: y, z from copies", 512: "A: x from a copy", 1024: "B copies explicit after A", 2048: "behind s_nop 3", 4096: "with truth path C"}


def main(trials=40):
    dev = torch.device("cuda:0")
    lib = _lib.load()
    probe = ctypes.CDLL(os.path.join(ROOT, "tools", "exp", "libpk_probe2.so"))
    probe.pk_probe2.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    big_np, _ = syn.kitti_batch(0, 16, 4096)
    xyz = torch.from_numpy(np.ascontiguousarray(big_np[:, :3, :].transpose(0, 2, 1))).to(dev)
    B, N = 16, 4096
    P = 1 << 19
    g = torch.Generator(device=dev).manual_seed(0)
    X96 = torch.randn(P, 96, device=dev, generator=g)
    W128 = torch.randn(128, 96, device=dev, generator=g)
    b128 = torch.randn(128, device=dev, generator=g)
    Y128 = torch.empty(P, 128, device=dev)
    aff96 = torch.zeros(4 * 96, device=dev); aff96[96:192] = 1; aff96[288:] = 1
    st128 = torch.zeros(8 * 2 * 128, device=dev, dtype=torch.float64)
    ws = torch.zeros(2 * (P // 128) * 128, device=dev)
    main_s = torch.cuda.current_stream().cuda_stream
    _lib.set_option("PN2_SPLIT_WG2", 0)

    def kern():
        assert lib.pn2_conv1x1_fwd_pool(p(X96), 96, p(aff96), p(W128), 96, p(b128), p(Y128), 128, P, 96, 128, p(st128), 128, p(b128), p(ws), None, main_s) == 0

    side = torch.cuda.Stream(device=dev)
    variants = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4, 8, 16, 32]
    for var in variants:
        out = torch.zeros(8, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for tr in range(trials):
            side.wait_stream(torch.cuda.current_stream())
            kern()
            with torch.cuda.stream(side):
                assert probe.pk_probe2(xyz.data_ptr(), B, N, 2048, var, out.data_ptr(), side.cuda_stream) == 0, var
            kern(); kern()
            torch.cuda.synchronize()
        o = out.cpu().numpy().view(np.uint32)
        label = ", ".join(v for k, v in BITS.items() if var & k) or "base (A packed, B scalar fenced, branches, read ahead, minima)"
        print("VAR %2d %-70s: %d workgroup runs; differing distances %d, minima %d, threads %d%s" % (var, label, o[3], o[0], o[1], o[2],
              ("; A (packed) != truth %d, B (scalar) != truth %d" % (o[4], o[5])) if var & 4096 else ""))


if __name__ == "__main__":
    main()
