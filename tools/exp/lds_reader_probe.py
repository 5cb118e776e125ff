#!/usr/bin/env python3
"""Round 6: tools/exp/lds_reader_probe.hip beside the pooled bf16-split forward: which first reader of a fresh wide LDS read sees stale lanes."""
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
    probe = ctypes.CDLL(os.path.join(ROOT, "tools", "exp", "liblds_reader_probe.so"))
    probe.reader_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
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

    def k_pool():
        assert lib.pn2_conv1x1_fwd_pool(p(X96), 96, p(aff96), p(W128), 96, p(b128), p(Y128), 128, P, 96, 128, p(st128), 128, p(b128), p(ws), None, main_s) == 0

    def k_none():
        pass

    sink = torch.zeros(256, device=dev)
    probe.spin.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    spin_names = {"bf16": (0, 3000, "a spinner of v_mfma_f32_32x32x16_bf16"), "f32": (1, 1500, "a spinner of v_mfma_f32_32x32x2_f32"),
                  "valu": (2, 60000, "a spinner of vector fmas (no MFMA)"), "bf16s": (3, 6000, "a spinner of v_mfma_f32_16x16x32_bf16"),
                  "bf16x4": (4, 1500, "v_mfma_f32_32x32x16_bf16, 4 independent accumulators"), "f32x4": (5, 750, "v_mfma_f32_32x32x2_f32, 4 independent accumulators"),
                  "f16x4": (6, 1500, "v_mfma_f32_32x32x16_f16, 4 independent accumulators"),
                  "f16s": (7, 6000, "a chain of v_mfma_f32_16x16x32_f16"), "f32s": (8, 6000, "a chain of v_mfma_f32_16x16x4_f32"),
                  "bf16sx4": (9, 3000, "v_mfma_f32_16x16x32_bf16, 4 independent accumulators")}

    def k_spin(kind, iters):
        def f():
            assert probe.spin(sink.data_ptr(), kind, 1024, iters, main_s) == 0
        return f

    kinds = {0: "b128, v_pk_add_f32 of the low pair", 1: "b128, two v_add_f32", 2: "b128, v_pk_mul_f32 of the high pair", 3: "b128, v_fma_f64 of the low pair",
             4: "b128, one idle state, v_pk_add_f32", 5: "b64, v_pk_add_f32", 6: "b128, v_mfma_f32_32x32x2_f32", 7: "b128, v_mfma_f32_32x32x16_bf16",
             8: "b128, pk add op_sel/neg (failing form)", 9: "b96, pk add op_sel/neg", 10: "b128, 4 VALU before wait, pk op_sel/neg",
             11: "b128, 4 VALU before wait, plain pk add", 12: "b96 onto its address reg, 4 VALU, pk",
             13: "b128, 16 VALU, wait, v_add_f32", 14: "b128, 16 VALU, wait, pk add", 15: "b128, 24 VALU, wait, v_add_f32",
             16: "b128, 16 VALU, wait, 1 idle, v_add_f32", 17: "b128, 16 VALU, wait, 4 idle, v_add_f32", 18: "2 x b32, 16 VALU, wait, v_add_f32",
             19: "b128, 8 VALU, wait, v_add_f32", 20: "b128, 12 VALU, wait, v_add_f32", 21: "b64, 16 VALU, wait, v_add_f32",
             22: "b96, wait, packed + 32-bit mix", 23: "b96, wait, 4 idle, packed + 32-bit mix", 24: "b96, wait, 32-bit only",
             25: "b96, wait, 16 idle, packed + 32-bit mix", 26: "b96 onto its address, wait, mix",
             27: "32 VALU writes under a b128 read", 28: "32 VALU writes under a b32 read", 29: "32 VALU writes, no LDS read", 30: "b128, wait, 32 VALU writes",
             31: "pk + 32-bit writes under a b128 read", 32: "pk + 32-bit writes, no LDS read", 33: "b128, wait, pk + 32-bit writes", 34: "b128, wait, 4 x (pk + 32-bit writes)",
             35: "TRUTH: b96, wait, idle, check", 36: "TRUTH: b96, wait, pk lo-half + pk hi-half readers", 37: "TRUTH: b96, wait, pk hi-half reader",
             38: "TRUTH: b96, wait, pk lo-half reader", 39: "TRUTH: b96, wait, both pk readers, check at once",
             40: "DENSE 16 x (pk HIGH-half read, copy of low reg)", 41: "DENSE 16 x (pk LOW-half read, copy of low reg)", 42: "DENSE HIGH, long after the read",
             43: "DENSE HIGH, pair written by v_mov (no LDS)", 44: "DENSE HIGH after ds_read_b64",
             45: "DENSE HIGH, no idles, address in dest, pair read to the end", 46: "DENSE HIGH x5, no idles", 47: "DENSE LOW x5, no idles",
             48: "RESULT: b96, 16 pk adds through the HIGH half", 49: "RESULT: b96, 16 pk adds through the LOW half", 50: "RESULT: pair by v_mov, HIGH half",
             51: "RESULT: b96, idle, HIGH half", 52: "RESULT: 2 x b32, HIGH half",
             53: "RESULT: plain v_pk_add (lo<-lo, hi<-hi)", 54: "RESULT: v_pk_add op_sel:[0,1], no neg", 55: "RESULT: v_pk_mul op_sel:[0,1]",
             56: "RESULT: v_pk_add, op_sel:[1,0] (src0)", 57: "RESULT: v_pk_add op_sel:[0,1] op_sel_hi:[1,0] (swap)", 58: "RESULT: v_pk_fma op_sel:[0,1,0]",
             60: "RESULT: v_pk_add, src1 an SGPR pair, op_sel:[0,1]", 61: "RESULT: v_pk_fma op_sel:[0,0,1] (src2)", 62: "RESULT: what the wrong lanes hold",
             63: "RESULT: v_pk_fma op_sel:[1,0,0] (src0 HIGH)", 64: "RESULT: v_pk_fma op_sel_hi:[0,1,1] (src0 LOW bcast)", 65: "RESULT: v_pk_fma op_sel_hi:[1,1,0] (src2 LOW bcast)",
             66: "RESULT: v_pk_mov_b32 op_sel:[1,0]"}
    co = None
    for a in list(sys.argv):
        if a.startswith("--co="):
            co = a[5:]
            sys.argv.remove(a)
    alone = "--alone" in sys.argv
    sys.argv = [a for a in sys.argv if a != "--alone"]
    if len(sys.argv) > 1:
        kinds = {k: v for k, v in kinds.items() if k in [int(a) for a in sys.argv[1:]]}
    side = torch.cuda.Stream(device=dev)
    for kind, label in kinds.items():
        runners = (("alone", k_none),) if alone else (("beside the pooled split forward", k_pool),)
        if co:
            runners = tuple(("beside " + spin_names[c][2], k_spin(spin_names[c][0], spin_names[c][1])) for c in co.split(","))
        for name, kern in runners:
            out = torch.zeros(8 + 8 * 32, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            for tr in range(trials):
                side.wait_stream(torch.cuda.current_stream())
                kern()
                with torch.cuda.stream(side):
                    assert probe.reader_probe(xyz.data_ptr(), B, N, 2048, kind, out.data_ptr(), side.cuda_stream) == 0
                kern(); kern()
                torch.cuda.synchronize()
            o = out.cpu().numpy().view(np.uint32)
            lanes = sorted({int(o[9 + 4 * k]) % 64 for k in range(min(32, int(o[2])))})
            if kind == 62 and o[2]:
                for k in range(min(4, int(o[2]))):
                    print("      results (low, high) = (%08x, %08x); wanted y = %08x in both; the pair's LOW dword x = %08x" % (o[136 + 4 * k], o[139 + 4 * k], o[137 + 4 * k], o[138 + 4 * k]))
            print("%-42s %-32s: %d workgroup runs; first reader != late reader %d times in %d threads%s" % (
                label + ",", name, o[3], o[0], o[2], ("; lanes seen " + str(lanes)) if lanes else ""))


if __name__ == "__main__":
    main()
