#!/usr/bin/env python3
"""Device time of the two section-8(f)3 kernels at benchmark sizes (HIP events on the launch stream).

    python tools/bench_train.py            # one JSON line per kernel
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnet12_amd import _lib, loader, optim      # noqa: E402


def timed(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3          # us


def main():
    dev = torch.device("cuda:0")
    _lib.load()
    for name, n in (("ssg_semseg", 968173), ("msg_semseg", 1735001), ("x16", 16 * 1735001)):
        p = [torch.nn.Parameter(torch.randn(n, device=dev))]
        opt = optim.Adam(p, lr=1e-3, weight_decay=1e-4)
        p[0].grad.normal_()
        us = timed(opt.step)
        ref_p = [torch.nn.Parameter(torch.randn(n, device=dev))]
        ref = torch.optim.Adam(ref_p, lr=1e-3, weight_decay=1e-4)
        ref_p[0].grad = torch.randn(n, device=dev)
        us_ref = timed(ref.step)
        print(json.dumps({"kernel": "pn2_adam_step", "case": name, "elements": n, "us": round(us, 2),
                          "GB/s": round(28.0 * n / us / 1e3, 1), "aten_single_tensor_us": round(us_ref, 2)}))
    rng = np.random.default_rng(0)
    for B, M, N in ((16, 20000, 4096), (8, 120000, 65536)):
        scans = [rng.uniform(-60, 60, (M, 4)).astype(np.float32) for _ in range(B)]
        labels = [rng.integers(0, 19, M).astype(np.int32) for _ in range(B)]
        store = loader.ScanStore(scans, labels, dev)
        gen = torch.Generator(device=dev)
        gen.manual_seed(0)
        whole = timed(lambda: loader.prepare_batch(store, list(range(B)), N, train=True, rng=gen), reps=20)
        with _lib.call_profile() as calls:
            for _ in range(20):
                loader.prepare_batch(store, list(range(B)), N, train=True, rng=gen)
            torch.cuda.synchronize()
        ker = float(np.median([a.elapsed_time(b) for _, _, a, b, _k in calls])) * 1e3
        # algorithmic bytes per output point: choice 8 + raw row 16 + noise row 16 + label 4 + out 16 + label out 8
        print(json.dumps({"kernel": "pn2_prepare_clouds", "B": B, "M": M, "N": N, "kernel_us": round(ker, 2),
                          "GB/s": round(68.0 * B * N / ker / 1e3, 1), "prepare_batch_device_rng_us": round(whole, 1)}))


if __name__ == "__main__":
    main()
