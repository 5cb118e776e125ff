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
    with _lib.call_profile() as calls:
        fwd()
        torch.cuda.synchronize()
        agg = {}
        for name, _, e0, e1 in calls:
            agg[name] = agg.get(name, 0.0) + e0.elapsed_time(e1)
    print(json.dumps({"metric": "single-cloud forward latency, PointNet2SemSeg(19, 1) eval", "points": args.points,
                      "eager_ms": round(eager_ms, 3), "graph_ms": None if graph_ms is None else round(graph_ms, 3),
                      "stream_ms_per_frame": None if stream_ms is None else round(stream_ms, 3),
                      "points_per_s": round(args.points / ((graph_ms or eager_ms) * 1e-3), 1),
                      "kernels_ms": {k: round(v, 4) for k, v in sorted(agg.items(), key=lambda kv: -kv[1])}}))


if __name__ == "__main__":
    main()
