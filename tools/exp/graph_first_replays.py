"""Gradient mass of the first replays of a captured step against the same steps run eagerly."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pointnet12_amd import parallel, pointnet2 as M, synthetic as syn
from pointnet12_amd.graph import GraphedStep
from pointnet12_amd.loss import nll_loss
dev = torch.device("cuda", 0)
B, N = int(os.environ.get("B", "2")), int(os.environ.get("N", "1024"))
for mode in ("eager", "graph", "graph+prefetch"):
    torch.manual_seed(0)
    net = M.PointNet2SemSeg(13, 6).to(dev).train()
    bucket = parallel.FlatGradBucket(net, direct=True)
    pts, lab = syn.kitti_batch(0, B, N)
    pts, lab = torch.from_numpy(pts).to(dev), torch.from_numpy(lab).to(dev)

    def step():
        bucket.zero()
        lp = net(pts)
        loss = nll_loss(lp.reshape(-1, 13), lab.reshape(-1))
        loss.backward()
        return loss
    torch.manual_seed(1)
    if mode == "eager":
        run = step
        for _ in range(int(os.environ.get("WARM", "2")) + 1):
            step()
    elif mode == "graph":
        run = GraphedStep(step, dev, warmup=int(os.environ.get("WARM", "2")))
    else:
        run = GraphedStep(step, dev, warmup=int(os.environ.get("WARM", "2")), geometry_fn=lambda: net.features(pts))
    out = []
    for it in range(8):
        loss = run()
        torch.cuda.synchronize()
        out.append((round(float(loss), 4), float(bucket.flat.double().abs().sum())))
    print(mode, ["%.4f/%.3g" % o for o in out])
