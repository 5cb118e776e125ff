import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch, torch.nn as nn
from pointnet12_amd import pointnet_util as U
from test_mlp_gpu import torch_mlp
dev = torch.device("cuda:0")
P, pool, chans = 131072, 32, [9, 32, 32, 64]
for mode in ("signed", "negonly", "zeroonly"):
    gen = torch.Generator().manual_seed(P + len(chans))
    c_in = chans[0]; ld = (c_in + 3) & ~3
    rows = torch.zeros(P, ld); rows[:, :c_in] = torch.randn(P, c_in, generator=gen) * 2 + 0.5
    convs = nn.ModuleList([nn.Conv2d(a, b, 1) for a, b in zip(chans[:-1], chans[1:])])
    bns = nn.ModuleList([nn.BatchNorm2d(b) for b in chans[1:]])
    for bn in bns:
        bn.weight.data.uniform_(-1.5, 1.5, generator=gen)
        if mode == "zeroonly":
            bn.weight.data.abs_()
        if mode != "negonly":
            bn.weight.data[::5] = 0.0
        bn.bias.data.uniform_(-1.0, 1.0, generator=gen)
    convs.to(dev), bns.to(dev)
    x = rows.to(dev).requires_grad_(True)
    out = U.shared_mlp(x, c_in, convs, bns, pool, True)
    gw = torch.randn(out.shape, generator=gen).to(dev)
    (out * gw).sum().backward()
    mine = [x.grad[:, :c_in].clone()] + [p.grad.clone() for p in list(convs.parameters()) + list(bns.parameters())]
    x64 = rows[:, :c_in].to(dev).double().requires_grad_(True)
    c64 = nn.ModuleList([nn.Conv2d(a, b, 1) for a, b in zip(chans[:-1], chans[1:])]).to(dev).double()
    b64 = nn.ModuleList([nn.BatchNorm2d(b) for b in chans[1:]]).to(dev).double()
    c64.load_state_dict({k: v.double() for k, v in convs.state_dict().items()})
    b64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in bns.state_dict().items()})
    ref = torch_mlp(x64, c64, b64, pool, True, torch.float64)
    (ref * gw.double()).sum().backward()
    print(mode, "fwd err", float((out.double() - ref).abs().max()))
    theirs = [x64.grad] + [p.grad for p in list(c64.parameters()) + list(b64.parameters())]
    names = ["x"] + ["conv." + n for n, _ in convs.named_parameters()] + ["bn." + n for n, _ in bns.named_parameters()]
    for n, a, b in zip(names, mine, theirs):
        e = (a.double() - b).abs()
        sc = max(float(b.abs().max()), 1e-9)
        bad = (e > 1e-3 * sc).nonzero()
        print("   %-14s max err %.3e of max %.3e; entries beyond 1e-3: %d %s" % (n, float(e.max()), sc, len(bad), bad[:6].flatten().tolist() if len(bad) else ""))
