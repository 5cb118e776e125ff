"""Which of the two fp32 evaluations of ONE stage is the outlier?  Re-runs tests/test_parity_stages_gpu.py's teacher-forced
stage on the GPU and the oracle stage in fp32 AND fp64, prints per-tensor relative L2 errors of every pair.

    python tools/exp/stage_debug.py ssg 16 4096 sa3 [sa4 ...]
"""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_parity_stages_gpu as S          # noqa: E402
from pointnet12_amd import synthetic as syn  # noqa: E402

kind, B, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
names = sys.argv[4:]
scale = 16 if (kind == "msg" and N == 65536) else 1
dev = torch.device("cuda:0")
pts_np, lab_np = syn.kitti_batch(3, B, N)
pts, labels = torch.from_numpy(pts_np), torch.from_numpy(lab_np)
net, orc = S._nets(kind, dev, scale)
pristine = copy.deepcopy(orc)
starts = S._fps_starts(orc, kind, B, N)
rec = S._oracle_pass(orc, kind, pts, labels)
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
for name in names:
    r = rec[name]
    res = {}
    for tag, dt in (("o32", torch.float32), ("o64", torch.float64)):
        o_in = [None if t is None else t.clone().to(dt).requires_grad_(g) for t, g in zip(r["in"], r["in_grad"])]
        mod = copy.deepcopy(getattr(pristine, name)).to(dt).train()
        kw = {"start": starts[name]} if name in starts else {}
        out = mod(*o_in, **kw)
        out = out if isinstance(out, tuple) else (out,)
        torch.autograd.backward([o for o, g in zip(out, r["gout"]) if g is not None], [g.to(dt) for g in r["gout"] if g is not None])
        res[tag] = (out[-1].detach(), [None if (t is None or t.grad is None) else t.grad for t in o_in], {k: p.grad for k, p in mod.named_parameters()})
    runs = []
    for rep in range(2):
        h_in = [None if t is None else t.to(dev).requires_grad_(g) for t, g in zip(r["in"], r["in_grad"])]
        net.zero_grad(set_to_none=True)
        h_mod = getattr(net, name)
        kw = {"fps_start": starts[name].to(dev)} if name in starts else {}
        h_out = h_mod(*h_in, **kw)
        h_out = h_out if isinstance(h_out, tuple) else (h_out,)
        torch.autograd.backward([o for o, g in zip(h_out, r["gout"]) if g is not None], [g.to(dev) for g in r["gout"] if g is not None])
        torch.cuda.synchronize()
        runs.append((h_out[-1].detach().cpu(), [None if (t is None or t.grad is None) else t.grad.cpu() for t in h_in],
                     {k: p.grad.detach().cpu().clone() for k, p in h_mod.named_parameters()}))
    hip, hip2 = runs
    print("==== %s %s: forward |hip-o32| %.3g |hip-o64| %.3g |o32-o64| %.3g" % (kind, name, float((hip[0] - res["o32"][0]).abs().max()),
          float((hip[0].double() - res["o64"][0]).abs().max()), float((res["o32"][0].double() - res["o64"][0]).abs().max())))
    for i, g in enumerate(hip[1]):
        if g is None:
            continue
        g32, g64 = res["o32"][1][i], res["o64"][1][i]
        sc = float(g64.abs().max())
        bad = lambda a, b: int(((a.double() - b.double()).abs().amax(dim=1) > 5e-5 * sc).sum())
        print("  in%d grad: L2 hip-o64 %.3g  o32-o64 %.3g  hip-o32 %.3g  hip-hip2 %.3g | rows beyond tol: hip-o64 %d o32-o64 %d hip-o32 %d of %d" % (
            i, rel(g, g64), rel(g32, g64), rel(g, g32), rel(g, hip2[1][i]), bad(g, g64), bad(g32, g64), bad(g, g32), g.shape[0] * g.shape[2]))
    for k in hip[2]:
        g, g32, g64 = hip[2][k], res["o32"][2][k], res["o64"][2][k]
        print("  %-28s L2 hip-o64 %.3g  o32-o64 %.3g  hip-o32 %.3g  hip-hip2 %.3g   |g64| %.3g" % (k, rel(g, g64), rel(g32, g64), rel(g, g32), rel(g, hip2[2][k]), float(g64.norm())))
