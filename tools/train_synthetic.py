#!/usr/bin/env python3
"""The reference's training loop (semseg.py:120-150) on synthetic scans, every step of it on the HIP library:
resident raw scans -> pn2_prepare_clouds -> PointNet2SemSeg forward -> nll_loss -> backward into the flat gradient
bucket -> (gradient all-reduce when launched under torch.distributed.run) -> pn2_adam_step, StepLR as semseg.py:113.

Labels are a function of the normalised height and intensity, so the loss must fall; the script prints the loss
curve and the all-inclusive throughput (loader + step + optimiser), which bench.py's metric deliberately excludes.

    python tools/train_synthetic.py --steps 200 --batch 16 --npoints 4096 [--msg] [--graph]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnet12_amd import graph, loader, optim, parallel, pointnet2, synthetic as syn   # noqa: E402
from pointnet12_amd.loss import nll_loss                                                  # noqa: E402

CLASSES = 13


def raw_scans(count, M):
    scans, labels = [], []
    for i in range(count):
        n = syn.kitti_cloud(syn.SEED_BASE + i, M, M, 1)[:, :4].astype(np.float32)         # normalised
        raw = np.stack([n[:, 0] * 70, n[:, 1] * 70, n[:, 2] * 3, n[:, 3] / 2 + 0.5], 1).astype(np.float32)
        height = np.clip(((n[:, 2] + 1) / 2 * 8).astype(np.int64), 0, 7)
        lab = np.where(n[:, 3] > 0.6, 8 + np.clip((height // 2), 0, 4), height)
        scans.append(raw)
        labels.append(lab.astype(np.int32))
    return scans, labels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--npoints", type=int, default=4096)
    ap.add_argument("--scans", type=int, default=64)
    ap.add_argument("--raw-points", type=int, default=20000)
    ap.add_argument("--msg", action="store_true")
    ap.add_argument("--graph", action="store_true", help="capture zero-grad + forward + loss + backward + Adam")
    ap.add_argument("--lr", type=float, default=1e-3)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    scans, labels = raw_scans(args.scans, args.raw_points)
    store = loader.ScanStore(scans, labels, dev)
    net = (pointnet2.PointNet2SemSegMsg if args.msg else pointnet2.PointNet2SemSeg)(CLASSES, feature_dims=1).to(dev)
    net.train()
    bucket = parallel.FlatGradBucket(net, direct=True)
    opt = optim.Adam(net.parameters(), lr=args.lr, betas=(0.9, 0.999), eps=1e-08, weight_decay=1e-4, bucket=bucket,
                     device_step=args.graph, fused_zero_grad=True)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=max(args.steps // 3, 1), gamma=0.5)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    pick = np.random.default_rng(0)
    pts = torch.empty(args.batch, args.npoints, 4, device=dev)
    lab = torch.empty(args.batch, args.npoints, device=dev, dtype=torch.int64)

    def compute():
        opt.zero_grad()                                            # free after the first step (fused into Adam)
        loss = nll_loss(net(pts.transpose(2, 1)).reshape(-1, CLASSES), lab.reshape(-1))
        loss.backward()
        bucket.all_reduce()
        opt.step()
        return loss.detach()

    def next_batch():
        loader.prepare_batch(store, pick.integers(0, len(store), args.batch), args.npoints, train=True, rng=gen,
                             out=(pts, lab))

    next_batch()
    step = graph.GraphedStep(compute, dev) if args.graph else compute      # (no geometry prefetch in this loop: the batch changes between replays)
    curve = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(args.steps):
        next_batch()
        loss = step()
        if it % max(args.steps // 20, 1) == 0 or it == args.steps - 1:
            curve.append((it, round(float(loss), 4)))              # the float() is this loop's only sync
        sched.step()
        if args.graph:
            opt.sync_lr()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"net": "msg" if args.msg else "ssg", "graph": args.graph, "steps": args.steps,
                      "batch": args.batch, "npoints": args.npoints, "ms_per_step_all_in": round(dt / args.steps * 1e3, 3),
                      "points_per_s_all_in": round(args.batch * args.npoints * args.steps / dt),
                      "loss_first": curve[0][1], "loss_last": curve[-1][1], "curve": curve}))


if __name__ == "__main__":
    main()
