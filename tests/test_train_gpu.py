"""GPU: the steps either side of the path through the C ABI -- pn2_adam_step against
torch.optim.Adam and the oracle, pn2_prepare_clouds bit-exact against the reference loader's golden vectors."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import train_ref as TR
from pointnet12_amd import loader, optim

pytestmark = pytest.mark.gpu

ADAM_TOL = 1e-6          # relative to max |param|; ATen fuses multiply-adds, the kernel does not (measured ~1e-7)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def flat(params):
    return torch.cat([p.detach().reshape(-1) for p in params]).cpu().numpy()


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


# ------------------------------------------------------------------------------------------------ loader
def test_prepare_clouds_golden_bit_exact(dev):
    g = golden("g8_train.npz")
    tags = list(g["loader_cases"])
    store = loader.ScanStore([g[t + "/raw"] for t in tags], [g[t + "/label"] for t in tags], dev)
    for i, tag in enumerate(tags):
        np.random.seed(int(g[tag + "/np_seed"]))
        pts, lab = loader.prepare_batch(store, [i], g[tag + "/points"].shape[0], train=bool(g[tag + "/train"]))
        assert pts.shape == (1,) + g[tag + "/points"].shape and lab.dtype == torch.int64
        assert (bits(pts[0].cpu().numpy()) == bits(g[tag + "/points"])).all(), tag
        assert (lab[0].cpu().numpy() == g[tag + "/labels"]).all(), tag


@pytest.mark.parametrize("train", [True, False])
def test_prepare_batch_matches_sequential_getitem(dev, train):
    """A batch draws cloud by cloud in the order a num_workers=0 DataLoader calls __getitem__."""
    rng = np.random.default_rng(5)
    scans = [np.concatenate([rng.uniform(-90, 90, (m, 2)), rng.uniform(-4, 4, (m, 1)), rng.uniform(-0.2, 1.2, (m, 1))],
                            1).astype(np.float32) for m in (1, 37, 4096, 9001)]
    labels = [rng.integers(0, 19, s.shape[0]).astype(np.int32) for s in scans]
    store = loader.ScanStore(scans, labels, dev)
    order = [3, 0, 2, 1, 3]
    np.random.seed(99)
    ref = [TR.prepare_cloud(scans[i], labels[i], 2048, train) for i in order]
    np.random.seed(99)
    pts, lab = loader.prepare_batch(store, order, 2048, train=train)
    for b in range(len(order)):
        assert (bits(pts[b].cpu().numpy()) == bits(ref[b][0])).all(), b
        assert (lab[b].cpu().numpy() == ref[b][1]).all(), b


def test_prepare_batch_device_generator_full_size(dev):
    """BASELINE cfg5 size (8 x 65 536 out of 120 000-point scans), device draws: every output row is a row of its
    own scan (intensity encodes the row number), labels follow, the jitter stays inside its clip."""
    B, M, N = 8, 120000, 65536
    rng = np.random.default_rng(1)
    scans, labels = [], []
    for b in range(B):
        s = rng.uniform(-60, 60, (M, 4)).astype(np.float32)
        s[:, 2] = rng.uniform(-2.5, 2.5, M)
        s[:, 3] = (np.arange(M) + 0.5) / M                          # row number, survives (i - 0.5) * 2 exactly enough
        scans.append(s)
        labels.append(((np.arange(M) * 7 + b) % 19).astype(np.int32))
    store = loader.ScanStore(scans, labels, dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(4)
    pts, lab = loader.prepare_batch(store, list(range(B)), N, train=False, rng=gen)
    p = pts.cpu().numpy()
    row = np.rint(((p[..., 3].astype(np.float64) / 2 + 0.5) * M) - 0.5).astype(np.int64)
    assert row.min() >= 0 and row.max() < M
    for b in range(B):
        assert (bits(p[b]) == bits(TR.normalize(scans[b])[row[b]])).all()
        assert (lab[b].cpu().numpy() == labels[b][row[b]]).all()
        assert len(np.unique(row[b])) > 0.35 * N                    # with replacement: ~42 % distinct expected
    gen.manual_seed(4)
    jit, _ = loader.prepare_batch(store, list(range(B)), N, train=True, rng=gen)
    d = (jit - pts).abs()
    assert float(d.max()) <= 0.05 + 1e-6 and float(d.mean()) > 0.005
    # duplicates of one raw row share their noise, as in the reference (jitter before resampling)
    r0, j0 = row[0], jit[0].cpu().numpy() - p[0]
    first = {}
    for n in range(4096):
        k = int(r0[n])
        if k in first:
            assert np.abs(j0[n] - j0[first[k]]).max() < 1e-6
        first[k] = n


def test_prepare_clouds_argument_checks(dev):
    from pointnet12_amd import _lib
    lib = _lib.load()
    assert lib.pn2_prepare_clouds(None, None, None, None, None, None, None, 1, 1, None, None, None, None) == -1
    with pytest.raises(ValueError):
        loader.ScanStore([np.zeros((4, 3), np.float32)], device=dev)
    with pytest.raises(ValueError):
        loader.ScanStore([np.zeros((4, 4), np.float32)], [np.zeros(3, np.int32)], device=dev)


# ------------------------------------------------------------------------------------------------ Adam
def test_adam_golden(dev):
    g = golden("g8_train.npz")
    shapes = [(64, 9, 1, 1), (64,), (13, 128, 1)]
    p0, off, params = g["adam/param0"], 0, []
    for s in shapes:
        n = int(np.prod(s))
        params.append(torch.nn.Parameter(torch.from_numpy(p0[off:off + n].reshape(s).copy()).to(dev)))
        off += n
    opt = optim.Adam(params, lr=1e-3, betas=(0.9, 0.999), eps=1e-08, weight_decay=float(g["adam/weight_decay"]))
    for t, (grad, lr, ref) in enumerate(zip(g["adam/grads"], g["adam/lr"], g["adam/after"]), 1):
        opt.zero_grad()
        off = 0
        for p in params:
            p.grad.add_(torch.from_numpy(grad[off:off + p.numel()]).to(dev).view_as(p))
            off += p.numel()
        opt.param_groups[0]["lr"] = float(lr)
        opt.step()
        assert rel(flat(params), ref) <= ADAM_TOL, t
    sd = opt.state_dict()["state"]
    assert float(sd[0]["step"]) == 12
    assert rel(torch.cat([sd[i]["exp_avg"].reshape(-1) for i in range(3)]).cpu().numpy(), g["adam/exp_avg"]) <= ADAM_TOL
    assert rel(torch.cat([sd[i]["exp_avg_sq"].reshape(-1) for i in range(3)]).cpu().numpy(),
               g["adam/exp_avg_sq"]) <= ADAM_TOL


@pytest.mark.parametrize("wd,sizes", [(0.0, [(7,), (3, 5), (1,)]), (1e-4, [(1000003,), (64, 9, 1, 1), (2,)])])
def test_adam_matches_torch_and_oracle(dev, wd, sizes):
    """Ragged sizes (total not a multiple of 4: the scalar tail), 8 steps, StepLR in the loop as semseg.py:113."""
    torch.manual_seed(11)
    init = [torch.randn(s) for s in sizes]
    ref_p = [torch.nn.Parameter(t.clone()) for t in init]
    my_p = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    ref = torch.optim.Adam(ref_p, lr=2e-3, betas=(0.9, 0.999), eps=1e-08, weight_decay=wd)
    mine = optim.Adam(my_p, lr=2e-3, betas=(0.9, 0.999), eps=1e-08, weight_decay=wd)
    s_ref = torch.optim.lr_scheduler.StepLR(ref, step_size=3, gamma=0.5)
    s_mine = torch.optim.lr_scheduler.StepLR(mine, step_size=3, gamma=0.5)
    o_p = flat(ref_p).copy()
    o_m, o_v = np.zeros_like(o_p), np.zeros_like(o_p)
    for t in range(1, 9):
        grads = [torch.randn(s) * 10.0 ** float(torch.randint(-5, 2, (1,))) for s in sizes]
        ref.zero_grad()
        mine.zero_grad()
        for p, q, gr in zip(ref_p, my_p, grads):
            p.grad = gr.clone()
            q.grad.copy_(gr)
        lr = ref.param_groups[0]["lr"]
        assert mine.param_groups[0]["lr"] == lr
        ref.step()
        mine.step()
        TR.adam_step(o_p, np.concatenate([gr.reshape(-1).numpy() for gr in grads]), o_m, o_v, t, lr=lr, weight_decay=wd)
        s_ref.step()
        s_mine.step()
        assert rel(flat(my_p), flat(ref_p)) <= ADAM_TOL, t
        assert rel(flat(my_p), o_p) <= ADAM_TOL, t
    for p, q in zip(ref_p, my_p):
        assert q.shape == p.shape and q.is_contiguous()


def test_adam_state_dict_round_trip_with_torch(dev):
    torch.manual_seed(2)
    init = [torch.randn(33, 5), torch.randn(33)]
    a = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    b = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    ta = torch.optim.Adam(a, lr=1e-3, weight_decay=1e-4)
    for _ in range(3):
        for p in a:
            p.grad = torch.randn_like(p)
        ta.step()
    mine = optim.Adam(b, lr=1e-3, weight_decay=1e-4)
    with torch.no_grad():
        for p, q in zip(a, b):
            q.copy_(p)
    mine.load_state_dict(ta.state_dict())                           # resume a torch.optim.Adam checkpoint
    assert mine.steps_taken() == [3]
    for _ in range(2):
        grads = [torch.randn_like(p) for p in a]
        mine.zero_grad()
        for p, q, gr in zip(a, b, grads):
            p.grad = gr
            q.grad.copy_(gr)
        ta.step()
        mine.step()
    assert rel(flat(b), flat(a)) <= ADAM_TOL
    tb = torch.optim.Adam([torch.nn.Parameter(t.clone().to(dev)) for t in init], lr=1e-3, weight_decay=1e-4)
    tb.load_state_dict(mine.state_dict())                           # and back
    assert float(tb.state_dict()["state"][0]["step"]) == 5


def test_adam_device_step_replays_from_a_graph(dev):
    """device_step: t and lr live in HBM, the captured launch advances t itself; fused_zero_grad clears the bucket."""
    torch.manual_seed(5)
    init = [torch.randn(257, 3), torch.randn(1025)]
    a = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    b = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    host = optim.Adam(a, lr=1e-3, weight_decay=1e-4)
    graphed = optim.Adam(b, lr=1e-3, weight_decay=1e-4, device_step=True, fused_zero_grad=True)
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            graphed.step()
    torch.cuda.synchronize()
    # the capture itself launches nothing: parameters and the step cell are untouched
    assert graphed.steps_taken() == [0] and rel(flat(b), flat(a)) == 0
    for t in range(1, 8):
        grads = [torch.randn_like(p) * 0.1 for p in a]
        host.zero_grad()
        for p, q, gr in zip(a, b, grads):
            p.grad.copy_(gr)
            assert float(q.grad.abs().max()) == 0               # cleared by the previous fused step
            q.grad.copy_(gr)
        if t == 4:
            host.param_groups[0]["lr"] = graphed.param_groups[0]["lr"] = 2.5e-4
            graphed.sync_lr()
        host.step()
        graph.replay()
        torch.cuda.synchronize()
        assert graphed.steps_taken() == [t]
        assert rel(flat(b), flat(a)) <= 1e-7, t                 # same kernel, same scalars: equal up to pow() on device
    assert float(graphed._flat[0]["step_dev"][1]) == 0              # ticket re-armed


def test_adam_keeps_the_network_wired(dev):
    """Re-pointing the parameters into the flat buffer must not change the network, gradients must land in the
    shared bucket, and three optimiser steps must lower the loss on a fixed batch."""
    from pointnet12_amd import pointnet2, synthetic as syn
    from pointnet12_amd.loss import nll_loss
    from pointnet12_amd.parallel import FlatGradBucket
    torch.manual_seed(0)
    net = pointnet2.PointNet2SemSeg(13, feature_dims=1).to(dev)
    pts_np, lab_np = syn.kitti_batch(0, 2, 1024, 4)
    pts, labels = torch.from_numpy(pts_np).to(dev), torch.from_numpy(lab_np).to(dev)
    net.eval()
    torch.manual_seed(1)
    before = net(pts).detach().clone()
    bucket = FlatGradBucket(net, direct=True)
    try:
        opt = optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-4, bucket=bucket)
        torch.manual_seed(1)
        assert torch.equal(net(pts), before)
        net.train()
        losses = []
        for _ in range(4):
            opt.zero_grad()
            torch.manual_seed(1)
            loss = nll_loss(net(pts).reshape(-1, 13), labels.reshape(-1))
            loss.backward()
            assert float(bucket.flat.abs().sum()) > 0
            for p in net.parameters():
                assert p.grad.data_ptr() >= bucket.flat.data_ptr()
            opt.step()
            losses.append(float(loss.detach()))
        assert losses[-1] < losses[0], losses
    finally:
        from pointnet12_amd import pointnet_util
        pointnet_util.set_direct_grad_accumulation(False)


def test_adam_two_parameter_groups(dev):
    """Per-group lr / weight_decay (torch.optim's param_groups API): one flat buffer and one launch per group."""
    torch.manual_seed(21)
    init = [torch.randn(50, 7), torch.randn(50), torch.randn(9, 3, 1)]
    a = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    b = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    spec = lambda ps: [{"params": ps[:2], "lr": 1e-3, "weight_decay": 1e-4}, {"params": ps[2:], "lr": 5e-3}]
    ref = torch.optim.Adam(spec(a), lr=1e-2, betas=(0.9, 0.999), eps=1e-08)
    mine = optim.Adam(spec(b), lr=1e-2, betas=(0.9, 0.999), eps=1e-08)
    assert len(mine.param_groups) == 2 and mine.param_groups[1]["weight_decay"] == 0
    for _ in range(5):
        grads = [torch.randn_like(p) for p in a]
        mine.zero_grad()
        for p, q, gr in zip(a, b, grads):
            p.grad = gr
            q.grad.copy_(gr)
        ref.step()
        mine.step()
    assert rel(flat(b), flat(a)) <= ADAM_TOL
    assert mine.steps_taken() == [5, 5]
    with pytest.raises(NotImplementedError):
        optim.Adam([torch.nn.Parameter(torch.zeros(3, device=dev))], amsgrad=True)
    with pytest.raises(Exception):
        optim.Adam([torch.nn.Parameter(torch.zeros(3))])            # CPU parameters: no fallback


def test_prepare_batch_without_labels_and_into_static_buffers(dev):
    rng = np.random.default_rng(8)
    scans = [rng.uniform(-50, 50, (m, 4)).astype(np.float32) for m in (300, 5000)]
    store = loader.ScanStore(scans, None, dev)
    np.random.seed(3)
    pts, lab = loader.prepare_batch(store, [1, 0], 512, train=False)
    assert lab is None and pts.shape == (2, 512, 4)
    np.random.seed(3)
    out = torch.full((2, 512, 4), 7.0, device=dev)
    again, _ = loader.prepare_batch(store, [1, 0], 512, train=False, out=(out, None))
    assert again.data_ptr() == out.data_ptr() and torch.equal(out, pts)
    np.random.seed(3)
    ref = [TR.prepare_cloud(scans[i], np.zeros(len(scans[i]), np.int32), 512, False)[0] for i in (1, 0)]
    assert (bits(pts.cpu().numpy()) == bits(np.stack(ref))).all()
    with pytest.raises(ValueError):
        loader.prepare_batch(store, [0], 512, out=(torch.empty(1, 512, 3, device=dev), None))
    with pytest.raises(ValueError):
        loader.prepare_batch(store, [], 512)


def test_held_channel_first_view_follows_prepare_batch_into_static_buffers(dev):
    """``prepare_batch(out=...)`` refills a static buffer through a raw pointer (no ``_version`` bump).  A
    channel-first view of that buffer held across steps must not get the PREVIOUS batch's memoised channel-last copy back."""
    from pointnet12_amd import pointnet_util as U
    rng = np.random.default_rng(9)
    scans = [rng.uniform(-50, 50, (m, 4)).astype(np.float32) for m in (700, 900)]
    store = loader.ScanStore(scans, None, dev)
    out = torch.zeros(2, 256, 4, device=dev)
    held = out.permute(0, 2, 1)[:, :3, :]                      # [B, 3, N] view created once, as a training loop would
    np.random.seed(1)
    loader.prepare_batch(store, [0, 1], 256, train=False, out=(out, None))
    a = U._channel_last(held, "xyz")
    assert torch.equal(a, out[:, :, :3])
    assert U._channel_last(held, "xyz") is a                   # unchanged data: the same copy (sa1 and fp1 share it)
    np.random.seed(2)
    loader.prepare_batch(store, [1, 0], 256, train=False, out=(out, None))
    b = U._channel_last(held, "xyz")
    assert torch.equal(b, out[:, :, :3]) and not torch.equal(a, b)


def test_kitti_files_to_device_batch(dev, tmp_path):
    """.bin / .label files -> ScanStore -> one evaluation batch: every row is a normalised row of the filtered scan."""
    import os
    from pointnet12_amd import kitti
    g = golden("g9_kitti.npz")
    fv, fl = os.path.join(tmp_path, "a.bin"), os.path.join(tmp_path, "a.label")
    g["bin"].tofile(fv)
    g["label"].tofile(fl)
    lmap = {int(k): int(v) for k, v in zip(g["map_keys"], g["map_values"])}
    store = kitti.load_scans([(fv, fl), (fv, fl)], lmap, "inview", dev)
    assert len(store) == 2 and int(store.row_count[0]) == len(g["inview/points"])
    np.random.seed(1)
    pts, lab = loader.prepare_batch(store, [0, 1], 4096, train=False)
    np.random.seed(1)
    for b in range(2):
        ref_p, ref_l, _, _ = TR.prepare_cloud(g["inview/points"], g["inview/labels"], 4096, False)
        assert (bits(pts[b].cpu().numpy()) == bits(ref_p)).all()
        assert (lab[b].cpu().numpy() == ref_l).all()
