#!/usr/bin/env python3
"""Round 6: the recorded geometry of a captured step against the eager step's AND against the oracle, repeated until they differ."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.nn.functional as F

from oracle import geometry as OG
from pointnet12_amd import graph as G_
from pointnet12_amd import parallel
from pointnet12_amd import pointnet2 as M
from pointnet12_amd import pointnet_util as U
from pointnet12_amd.graph import GraphedStep


def main(reps=12, steps=6):
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(ROOT, "tests", "golden", "g6_nets.npz"))
    pts = torch.from_numpy(g["points"]).to(dev)
    labels = torch.from_numpy(g["labels"]).to(dev)
    xyz_np = np.ascontiguousarray(g["points"][:, :3, :].transpose(0, 2, 1))
    for rep in range(reps):
        out = {}
        for mode in ("eager", "prefetch"):
            torch.manual_seed(int(g["init_seed"]))
            net = M.PointNet2SemSegMsg(13, 6)
            net.drop1.p = 0.0
            net.to(dev).train()
            bucket = parallel.FlatGradBucket(net)

            def compute():
                bucket.zero()
                lp = net(pts)
                loss = F.nll_loss(lp.reshape(-1, 13), labels.reshape(-1))
                loss.backward()
                return loss
            torch.manual_seed(31)
            geos = []
            if mode == "eager":
                for _ in range(2):
                    compute()
                for r in range(steps):
                    state = torch.get_rng_state()
                    tape = U.GeometryTape()
                    U.set_geometry_tape(tape)
                    with torch.no_grad():
                        net.features(pts)
                    U.set_geometry_tape(None)
                    torch.set_rng_state(state)
                    geos.append([t.detach().cpu().clone() for t in G_._flatten(tape.items)])
                    float(compute())
            else:
                step = GraphedStep(compute, dev, warmup=2, geometry_fn=lambda: net.features(pts))
                for r in range(steps):
                    float(step())
                    geos.append([t.detach().cpu().clone() for t in G_._flatten(step._tapes[r % 2].items)])
            out[mode] = geos
        bad = []
        for r in range(steps):
            for i, (u, v) in enumerate(zip(out["eager"][r], out["prefetch"][r])):
                if u.dtype != torch.int32 and not torch.equal(u, v):
                    bad.append((r, i))
        print("rep %d: differing (step, item): %s" % (rep, bad))
        for r, i in bad[:2]:
            if i != 0:
                continue
            e, p = out["eager"][r][0].numpy(), out["prefetch"][r][0].numpy()
            ref_e = OG.farthest_point_sample(xyz_np, e.shape[1], e[:, 0].copy())
            ref_p = OG.farthest_point_sample(xyz_np, p.shape[1], p[:, 0].copy())
            first = int(np.argmax((e != p).any(0)))
            print("   step %d: starts eager %s prefetch %s; first differing column %d; eager == oracle: %s; prefetch == oracle: %s" %
                  (r, e[:, 0], p[:, 0], first, bool((e == ref_e).all()), bool((p == ref_p).all())))
            print("   eager    ", e[:, max(0, first - 2):first + 4].tolist())
            print("   prefetch ", p[:, max(0, first - 2):first + 4].tolist())
            print("   oracle(p)", ref_p[:, max(0, first - 2):first + 4].tolist())


if __name__ == "__main__":
    main()
