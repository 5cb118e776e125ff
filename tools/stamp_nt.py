#!/usr/bin/env python3
"""Diagnostic: where an NT GEMM workgroup spends its cycles (needs `make -C pointnet12_amd/csrc STAMP=1`)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointnet12_amd import _lib
from pointnet12_amd._lib import ptr as p
lib = _lib.load(); raw = ctypes.CDLL(_lib.LIB_PATH)
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
names = ["lds_store+fetch(vmcnt)", "barrier_k", "mfma_loop", "barrier_epi", "stage_write", "barrier_stage", "readback+store", "barrier_end"]
for P, K, N in [(1048576, 96, 128), (262144, 196, 256), (1048576, 64, 96)]:
    X = torch.randn(P, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    Y = torch.empty(P, N, device=dev); stats = torch.zeros(8 * 2 * N, device=dev, dtype=torch.float64)
    aff = torch.ones(4 * K, device=dev)
    for _ in range(3):
        lib.pn2_conv1x1_fwd(p(X), K, p(aff), p(W), K, p(b), p(Y), N, P, K, N, p(stats), None, None, st)
    buf = (ctypes.c_ulonglong * (8 * 512))()
    raw.pn2_debug_stamps(buf, 8 * 512)
    a = np.array(buf, dtype=np.float64).reshape(512, 8)
    a = a[a.sum(1) > 0]
    tot = a.sum(1).mean()
    print((P, K, N), "blocks", len(a), "cycles/block %.0f" % tot)
    for n, v in zip(names, a.mean(0)):
        print("   %-24s %6.1f %%" % (n, 100 * v / tot))
