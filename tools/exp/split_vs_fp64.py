"""Per gradient tensor: error against the fp64 evaluation of the fp32-pipe kernels (PN2_SPLIT=0), of the bf16-split kernels, and of
plain torch fp32 on the GPU -- median and 99.5th percentile of |err| / max|ref|."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, torch.nn as nn
from pointnet12_amd import _lib, pointnet_util as U
from tests.test_mlp_gpu import torch_mlp

dev = torch.device("cuda:0")
for P, pool, chans in [(262144, 128, [9, 64, 96, 128]), (262144, 0, [128, 128, 128, 64]), (131072 + 64, 0, [12, 64, 64, 96])]:
    gen = torch.Generator().manual_seed(P + pool)
    c_in = chans[0]
    rows = torch.zeros(P, (c_in + 3) & ~3)
    rows[:, :c_in] = torch.randn(P, c_in, generator=gen) * 2 + 0.5
    convs = nn.ModuleList([nn.Conv2d(a, b, 1) for a, b in zip(chans[:-1], chans[1:])]).to(dev)
    bns = nn.ModuleList([nn.BatchNorm2d(b) for b in chans[1:]]).to(dev)
    params = list(convs.parameters()) + list(bns.parameters())
    names = ["x"] + [n for n, _ in list(convs.named_parameters()) + list(bns.named_parameters())]
    gw = None
    res = {}
    for arm, (sp, sr) in {"fp32 pipe": (0, 0), "bf16 split": (1, 2)}.items():
        _lib.set_option("PN2_SPLIT", sp); _lib.set_option("PN2_SPLIT_RES", sr)
        for bn in bns:
            bn.reset_running_stats()
        x = rows.to(dev).requires_grad_(True)
        out = U.shared_mlp(x, c_in, convs, bns, pool, True)
        if gw is None:
            gw = torch.randn(out.shape, generator=gen).to(dev)
        g = torch.autograd.grad((out * gw).sum(), [x] + params)
        res[arm] = [g[0][:, :c_in]] + list(g[1:])
    _lib.set_option("PN2_SPLIT", 1); _lib.set_option("PN2_SPLIT_RES", 1)
    x64 = rows[:, :c_in].to(dev).double().requires_grad_(True)
    c64 = nn.ModuleList([nn.Conv2d(a, b, 1) for a, b in zip(chans[:-1], chans[1:])]).to(dev).double()
    b64 = nn.ModuleList([nn.BatchNorm2d(b) for b in chans[1:]]).to(dev).double()
    c64.load_state_dict({k: v.double() for k, v in convs.state_dict().items()})
    b64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in bns.state_dict().items()})
    ref = torch.autograd.grad((torch_mlp(x64, c64, b64, pool, True, torch.float64) * gw.double()).sum(), [x64] + list(c64.parameters()) + list(b64.parameters()))
    x32 = rows[:, :c_in].to(dev).requires_grad_(True)
    c32 = nn.ModuleList([nn.Conv2d(a, b, 1) for a, b in zip(chans[:-1], chans[1:])]).to(dev); c32.load_state_dict(convs.state_dict())
    b32 = nn.ModuleList([nn.BatchNorm2d(b) for b in chans[1:]]).to(dev); b32.load_state_dict(bns.state_dict())
    for bn in b32:
        bn.reset_running_stats()
    res["torch fp32"] = torch.autograd.grad((torch_mlp(x32, c32, b32, pool, True, torch.float32) * gw).sum(), [x32] + list(c32.parameters()) + list(b32.parameters()))
    print("P=%d pool=%d chans=%s   (median | 99.5th percentile of |err| / max|ref|)" % (P, pool, chans))
    for i, n in enumerate(names):
        r = ref[i]
        scale = max(float(r.abs().max()), 1e-12)
        if scale < 1e-9:
            continue
        line = "  %-10s" % n
        for arm in ("fp32 pipe", "bf16 split", "torch fp32"):
            e = (res[arm][i].double() - r).abs().flatten() / scale
            k = max(1, int(e.numel() * 0.995))
            line += "  %s %.1e | %.1e" % (arm, float(e.median()), float(e.kthvalue(k)[0]))
        print(line)
