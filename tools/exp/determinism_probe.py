#!/usr/bin/env python3
"""Round 6: which tensors of an MSG / SSG training step differ from run to run (eager, same batch, same FPS starts)?

python tools/exp/determinism_probe.py [msg|ssg] [runs]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from pointnet12_amd import pointnet_util as U
from pointnet12_amd import synthetic as syn
from pointnet12_amd.loss import nll_loss


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "msg"
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dev = torch.device("cuda:0")
    net = bench.build_net(workload, dev)
    pts_np, lab_np = syn.kitti_batch(0, 16, 4096)
    pts, labels = torch.from_numpy(pts_np).to(dev), torch.from_numpy(lab_np).to(dev)
    names = [n for n, _ in net.named_parameters()]
    outs = []
    acts = {}

    def hook(name):
        def f(m, i, o):
            t = o[1] if isinstance(o, tuple) else o
            acts.setdefault(name, []).append(t.detach().clone())
        return f
    for name, mod in net.named_children():
        mod.register_forward_hook(hook(name))
    for r in range(runs):
        torch.manual_seed(123)                      # the FPS start draw
        U.seed_fps_starts(123) if hasattr(U, "seed_fps_starts") else None
        net.zero_grad(set_to_none=True)
        lp = net(pts)
        loss = nll_loss(lp.reshape(-1, lp.shape[-1]), labels.reshape(-1))
        loss.backward()
        torch.cuda.synchronize()
        outs.append((loss.detach().clone(), lp.detach().clone(), [p.grad.detach().clone() for p in net.parameters()]))
    base = outs[0]
    print("%s, %d runs against the first:" % (workload, runs))
    for name, lst in acts.items():
        d = [int((lst[0] != t).sum()) for t in lst[1:]]
        print("  forward output of %-8s: elements that differ %s of %d" % (name, d, lst[0].numel()))
    print("  loss bits equal: %s (%s)" % ([bool(torch.equal(base[0], o[0])) for o in outs[1:]], [float(o[0]) for o in outs]))
    print("  log-probabilities: elements that differ %s" % [int((base[1] != o[1]).sum()) for o in outs[1:]])
    for k, n in enumerate(names):
        d = [int((base[2][k] != o[2][k]).sum()) for o in outs[1:]]
        rel = max(float((base[2][k] - o[2][k]).abs().max() / base[2][k].abs().max().clamp_min(1e-30)) for o in outs[1:])
        if any(d):
            print("  grad %-40s %6d elements: differ %s   max |diff| / max |g| %.1e" % (n, base[2][k].numel(), d, rel))
    same = [n for k, n in enumerate(names) if all(torch.equal(base[2][k], o[2][k]) for o in outs[1:])]
    print("  gradients bit-identical in all runs: %d of %d parameters" % (len(same), len(names)))


if __name__ == "__main__":
    main()
