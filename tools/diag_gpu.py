import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, torch.nn as nn, torch.nn.functional as F
from pointnet12_amd import pointnet_util as U, pointnet2 as M
from test_mlp_gpu import torch_mlp
dev = torch.device("cuda:0")
g = np.load("tests/golden/g6_nets.npz")
for tag, make in [("ssg", lambda: M.PointNet2SemSeg(13, 6)), ("msg", lambda: M.PointNet2SemSegMsg(13, 6))]:
    torch.manual_seed(int(g["init_seed"])); net = make(); net.drop1.p = 0.0; net.to(dev).train()
    pts = torch.from_numpy(g["points"]).to(dev); labels = torch.from_numpy(g["labels"]).to(dev)
    torch.manual_seed(int(g["fwd_seed"])); lp = net(pts); loss = F.nll_loss(lp.reshape(-1, 13), labels.reshape(-1)); loss.backward()
    print(tag, "lp err", np.abs(lp.detach().cpu().numpy() - g[tag + "/log_probs"]).max(), "loss", float(loss), float(g[tag + "/loss"]))
    grads = dict(net.named_parameters())
    for n, l2 in zip(g[tag + "/grad_names"], g[tag + "/grad_l2"]):
        n = str(n); mine = np.linalg.norm(grads[n].grad.double().cpu().numpy())
        if abs(mine - l2) > 5e-4 * l2: print("   ", n, mine, l2, abs(mine-l2)/l2)
# mlp case
P, pool, chans = 6400, 200, [515, 256, 512, 1024]
gen = torch.Generator().manual_seed(P + len(chans)); c_in = chans[0]; ld = (c_in + 3) & ~3
rows = torch.zeros(P, ld); rows[:, :c_in] = torch.randn(P, c_in, generator=gen) * 2 + 0.5
convs = nn.ModuleList([nn.Conv2d(a, b, 1) for a, b in zip(chans[:-1], chans[1:])]); bns = nn.ModuleList([nn.BatchNorm2d(b) for b in chans[1:]])
convs.to(dev), bns.to(dev)
x = rows.to(dev).requires_grad_(True)
out = U.shared_mlp(x, c_in, convs, bns, pool, True)
gw = torch.randn(out.shape, generator=gen).to(dev)
(out * gw).sum().backward()
gx = x.grad[:, :c_in].clone(); gp = [p.grad.clone() for p in list(convs.parameters()) + list(bns.parameters())]
for p in list(convs.parameters()) + list(bns.parameters()): p.grad = None
x32 = rows[:, :c_in].to(dev).requires_grad_(True)
for bn in bns: bn.reset_running_stats()
ref = torch_mlp(x32, convs, bns, pool, True, torch.float32)
(ref * gw).sum().backward()
print("fwd err vs torch32", float((out - ref).abs().max()))
e = (gx - x32.grad).abs(); print("gx err max", float(e.max()), "scale", float(x32.grad.abs().max()), "n>1e-4:", int((e > 1e-4).sum()), "of", e.numel())
bad = (e > 1e-4).nonzero(); print("bad rows", torch.unique(bad[:, 0])[:20].tolist())
for (n, p), a in zip(list(convs.named_parameters()) + list(bns.named_parameters()), gp):
    print(n, float((a - p.grad).abs().max()), float(p.grad.abs().max()))
