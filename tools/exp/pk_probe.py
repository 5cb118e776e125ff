#!/usr/bin/env python3
"""Round 6: tools/exp/pk_probe.hip (packed vs scalar fp32 distances in one thread) beside the pooled bf16-split forward.

Build first: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -shared -fPIC -o tools/exp/libpk_probe.so tools/exp/pk_probe.hip
Results of round 6: HISTORY.md, "pn2_fps beside the pooled bf16-split forward"."""
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


def main(trials=60):
    dev = torch.device("cuda:0")
    lib = _lib.load()
    probe = ctypes.CDLL(os.path.join(ROOT, "tools", "exp", "libpk_probe.so"))
    probe.pk_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
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
    A_ = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
    B_ = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)

    def k_pool():
        assert lib.pn2_conv1x1_fwd_pool(p(X96), 96, p(aff96), p(W128), 96, p(b128), p(Y128), 128, P, 96, 128, p(st128), 128, p(b128), p(ws), None, main_s) == 0

    def k_bf16_mm():
        torch.mm(A_, B_)

    sink = torch.zeros(256, device=dev)
    probe.lds_hammer.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]

    def k_hammer():
        assert probe.lds_hammer(sink.data_ptr(), 1024, 1500, main_s) == 0

    def k_none():
        pass

    side = torch.cuda.Stream(device=dev)
    modes = {0: "global memory", 1: "LDS ds_read_b96", 2: "LDS b96, wait + 16 idle states", 3: "LDS b96, lane 0's copy via SGPRs", 4: "LDS 3 x ds_read_b32",
             5: "LDS ds_read_b128", 6: "LDS, only lane 0 reads, SGPRs", 7: "LDS b96, wait + 4 idle states",
             8: "LDS b96, wait + 1 idle state", 9: "LDS b96, wait + 2 idle states", 10: "LDS b96, the wait in asm, 0 idle",
             11: "global memory -> VGPRs", 12: "LDS b96 issued an iteration ahead"}
    for use_lds in (1, 5, 12, 10, 8, 9, 7, 2, 4, 3, 6, 11, 0):
        for name, kern in (("beside the pooled split forward (1 WG/CU)", k_pool), ):
            out = torch.zeros(8 + 8 * 32, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            for tr in range(trials):
                side.wait_stream(torch.cuda.current_stream())
                kern()
                with torch.cuda.stream(side):
                    assert probe.pk_probe(xyz.data_ptr(), B, N, 2048, use_lds, out.data_ptr(), side.cuda_stream) == 0
                kern(); kern()
                torch.cuda.synchronize()
            o = out.cpu().numpy().view(np.uint32)
            print("centre from %-34s %-42s: %d workgroup runs; packed != scalar distances %d, running minima that ended different %d, threads affected %d" % (
                modes[use_lds] + ",", name, o[3], o[0], o[1], o[2]))
            pts_np = xyz.cpu().numpy()
            for k in range(min(2, int(o[2]))):
                bb, tt, slot = int(o[8 + 4 * k]), int(o[9 + 4 * k]), int(o[11 + 4 * k])
                dp, ds = o[136 + 4 * k: 137 + 4 * k].view(np.float32)[0], o[137 + 4 * k: 138 + 4 * k].view(np.float32)[0]
                far, prev = int(o[138 + 4 * k]), int(np.int32(o[139 + 4 * k]))
                q = pts_np[bb, (tt + slot * 512) % N]
                cur, old = pts_np[bb, far], pts_np[bb, prev] if prev >= 0 else pts_np[bb, far]
                combos = {}
                for mask in range(8):
                    c = np.float32([old[a] if (mask >> a) & 1 else cur[a] for a in range(3)])
                    d = q - c
                    combos[mask] = np.float32(np.float32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
                which_p = [m for m, v in combos.items() if v == dp]
                which_s = [m for m, v in combos.items() if v == ds]
                print("      lane %2d slot %d it %d: packed %.9g = centre with STALE axes mask %s; scalar %.9g = mask %s   (mask bit a set: axis a from the previous centre)" % (
                    tt % 64, slot, int(o[10 + 4 * k]), dp, which_p, ds, which_s))
            for k in range(0):
                print("      cloud %d thread %d (lane %d): first at iteration %d slot %d" % (o[8 + 4 * k], o[9 + 4 * k], o[9 + 4 * k] % 64, o[10 + 4 * k], o[11 + 4 * k]))


if __name__ == "__main__":
    main()
