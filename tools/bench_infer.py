#!/usr/bin/env python3
"""Single-cloud inference latency, the shape of the reference's viewer loop (pcdvis.py:113-136 / model/utils.py:15-34):
PointNet2SemSeg(19 classes, 1 feature channel) in eval mode on one [1, 4, N] cloud, N ~ 25 000, forward only.

    python tools/bench_infer.py [--points 25000] [--reps 50] [--no-graph]

Prints one JSON line: ms per cloud (eager launches and hipGraph replay: the LATENCY of one frame), ms per frame of a frame
stream with the next frame's geometry prefetched inside the same graph (throughput), and the per-entry-point device time.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pointnet12_amd import _lib
from pointnet12_amd import pointnet2 as M
from pointnet12_amd import pointnet_util as U
from pointnet12_amd import synthetic as syn
from pointnet12_amd.graph import FpsStartFeed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=25000)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = M.PointNet2SemSeg(19, 1).to(dev).eval()
    cloud, _ = syn.kitti_batch(7, 1, args.points, channels=4)          # xyz + intensity, as the shipped checkpoint expects
    pts = torch.from_numpy(cloud).to(dev)

    def fwd():
        with torch.no_grad():
            return net(pts)

    def timeit(fn, reps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    torch.manual_seed(1)
    eager_ms = timeit(fwd, args.reps)
    graph_ms = None
    if not args.no_graph:
        feed = FpsStartFeed(dev)
        g = torch.cuda.CUDAGraph()
        U.set_fps_start_feed(feed)
        U.set_capture_scope(object())
        try:
            with torch.cuda.graph(g):
                out = fwd()
        finally:
            U.set_capture_scope(None)
            U.set_fps_start_feed(None)

        def replay():
            feed.stage()
            g.replay()
        graph_ms = timeit(replay, args.reps)
    # a STREAM of frames (the viewer loop processes consecutive scans): the next frame's geometry -- its FPS chain above all,
    # one workgroup on one CU -- runs on a side branch of the same graph while this frame's MLP kernels use the chip
    # (pointnet12_amd.graph.GraphedStep, the training step's prefetch): time per frame, not the latency of one frame
    stream_ms = None
    if not args.no_graph:
        from pointnet12_amd.graph import GraphedStep
        torch.manual_seed(2)
        streamed = GraphedStep(fwd, dev, warmup=1, geometry_fn=lambda: net.features(pts))
        stream_ms = timeit(streamed, args.reps)
    import ctypes
    with _lib.call_profile() as calls:
        fwd()
        torch.cuda.synchronize()
        agg = {}
        for name, _, e0, e1, _k in calls:
            agg[name] = agg.get(name, 0.0) + e0.elapsed_time(e1)
        # ---- price every pn2_fused_eval launch ("closest to the ALG_BYTES_MIN roofline").  Algorithmic bytes
        # in the ALG_BYTES_MIN sense: what must cross HBM once -- inputs (rows, or xyz + features + centres + neighbour index),
        # pooled outputs, folded weights; the activations never leave LDS.  The weights are re-read by every 32-row tile and
        # are served by the L2: priced separately against the L2's 34.5 TB/s (MI355X_MICROARCH.md).
        arrays = {ctypes.addressof(v["arr"]): v["arr"] for v in U._fold_cache.values()}
        fused = []
        for name, a, e0, e1, _k in calls:
            if name != "pn2_fused_eval":
                continue
            X, ldx, B, N, S, Knb, D, layers, L, pool, ldo = a[0], a[1], a[6], a[7], a[8], a[9], a[10], a[12], a[13], a[14], a[16]
            arr = arrays.get(layers.value if hasattr(layers, "value") else layers)
            dims = [(arr[l].K, arr[l].N, arr[l].ldw) for l in range(L)] if arr is not None else []
            P = B if X is not None else B * S * Knb
            w_bytes = sum(4 * (n * ldw + n) for _, n, ldw in dims)
            in_bytes = 4 * P * ldx if X is not None else 4 * B * N * (3 + D) + 12 * B * S + 8 * B * S * Knb
            out_bytes = 4 * (P // pool if pool else P) * ldo
            flops = 2.0 * P * sum(k * n for k, n, _ in dims)
            us = e0.elapsed_time(e1) * 1e3
            alg = in_bytes + out_bytes + w_bytes
            l2w = (P // 32) * w_bytes
            fused.append({"rows": P, "layers": "->".join([str(dims[0][0])] + [str(n) for _, n, _ in dims]) if dims else "?",
                          "pool": pool, "us": round(us, 1), "alg_bytes_min": alg, "hbm_GBs": round(alg / us / 1e3, 1),
                          "hbm_frac": round(alg / us / 1e3 / 8000.0, 4), "l2_weight_bytes": l2w, "l2_GBs": round(l2w / us / 1e3, 1),
                          "l2_frac": round(l2w / us / 1e3 / 34500.0, 4), "TFLOPs": round(flops / us / 1e6, 2),
                          "mfma_frac": round(flops / us / 1e6 / 157.3, 4)})
    print(json.dumps({"metric": "single-cloud forward latency, PointNet2SemSeg(19, 1) eval", "points": args.points,
                      "eager_ms": round(eager_ms, 3), "graph_ms": None if graph_ms is None else round(graph_ms, 3),
                      "stream_ms_per_frame": None if stream_ms is None else round(stream_ms, 3),
                      "points_per_s": round(args.points / ((graph_ms or eager_ms) * 1e-3), 1),
                      "kernels_ms": {k: round(v, 4) for k, v in sorted(agg.items(), key=lambda kv: -kv[1])},
                      "fused_eval_launches": fused}))


if __name__ == "__main__":
    main()
