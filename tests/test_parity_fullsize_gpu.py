"""GPU: whole networks at BASELINE.json's full sizes against the oracle, with an fp64 yardstick.

cfg3 = SemSeg B=16 x 4096 x (3+6), SSG (the reference's own class, model/pointnet2.py:141-176) and MSG;
cfg5 = dense 65 536-point scans.  The oracle (oracle/torch_ref.py, pinned to the reference by tools/make_golden.py)
runs on the host CPU with the same parameters, inputs and FPS start draws, once in fp32 -- the reference's own
arithmetic -- and once in fp64 with the fp32 geometry (indices, 3-NN weights) held fixed: that second run is the
yardstick.  A nine-stage stack of training-mode BatchNorms amplifies fp32 rounding: the reference's arithmetic sits
5e-6 (median) to 7e-5 (max over 852 k log-probs) from the fp64 evaluation at B=16 x 4096, and moves 9e-5 against ITSELF
at B=2 x 1024 when only its thread count changes (tests/golden/g6_noise.npz, measured with the reference).  No fp32
implementation can therefore be asked for 1e-5 against the reference at network level (the modules are: see
test_modules_gpu.py, 1e-5 against the reference's numbers); what is asserted here is that the HIP path is as close
to the truth as the reference is:  error(HIP, fp64) <= FACTOR x error(oracle fp32, fp64), for log-probs (max and rms)
and per-tensor gradients, plus an absolute cap on |HIP - oracle fp32|.
Measured numbers are written to gpurun_out/parity_fullsize.json (copied to profiles/ per round).
"""
import copy
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, golden
from oracle import geometry as G
from oracle import torch_ref as T
from pointnet12_amd import pointnet2 as M
from pointnet12_amd import pointnet_util as U
from pointnet12_amd import synthetic as syn
from pointnet12_amd.loss import nll_loss

pytestmark = pytest.mark.gpu

FACTOR = 2.0            # log-probs: HIP at most this many times further from fp64 than the reference's arithmetic is
GRAD_FACTOR = 3.0       # gradients: dominated by the handful of argmax / ReLU decisions that fall the other way than in fp64
                        # (one flip at a pooled stage re-routes an O(1) gradient through every layer below it): the same
                        # build measured 0.4x and 2.2x the reference's distance on two runs that differ only in the order
                        # of the fp64 statistics atomics (profiles/r02_parity_fullsize.json; the six cases of that file sit at
                        # 0.5 .. 1.25x: round 4 lowered the factor from 4 to 3 and the worst-tensor allowance from 8 to 4.5)
REL_CAP = 4e-5          # |HIP - oracle fp32| on log-probs, relative to their magnitude, whatever the yardstick says
REPORT = {}


def _report(key, value):
    REPORT[key] = value
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_fullsize.json"), "w") as f:
            json.dump(REPORT, f, indent=1, sort_keys=True)
    except OSError:
        pass


def _zero_grad_bias(n):
    """conv / fc biases in front of a training-mode BatchNorm: the exact gradient is 0, what is there is rounding noise."""
    return (("conv" in n and n.endswith("bias") and n != "conv2.bias") or
            (n.startswith("fc") and n.endswith("bias") and n != "fc3.bias"))


def _nets(kind, dev, npoint_scale=1, seed=0):
    torch.manual_seed(seed)
    if kind == "ssg":
        orc = T.RefSSGSemSeg(13, 6, dropout=0.0)
        net = M.PointNet2SemSeg(13, 6)
    else:
        orc = T.RefMSGSemSeg(13, 6, dropout=0.0, npoint_scale=npoint_scale)
        net = M.PointNet2SemSegMsg(13, 6, npoint_scale=npoint_scale)
    net.load_state_dict(orc.state_dict())
    net.drop1.p = 0.0
    return net.to(dev).train(), orc.train()


def _run_oracle(orc, pts, labels, dtype):
    net = copy.deepcopy(orc).to(dtype)
    torch.manual_seed(1)                       # FPS start draws: CPU generator, sa1 first (pointnet_util.py:75)
    lp = net(pts.to(dtype))
    T.seg_loss(lp, labels).backward()
    return lp.detach().double(), {n: p.grad.detach().double() for n, p in net.named_parameters()}


def _run_hip(net, pts, labels, dev):
    net.zero_grad(set_to_none=True)
    torch.manual_seed(1)
    lp = net(pts.to(dev))
    loss = nll_loss(lp.reshape(-1, lp.shape[-1]), labels.to(dev).reshape(-1))
    loss.backward()
    torch.cuda.synchronize()
    return lp.detach().double().cpu(), {n: p.grad.detach().double().cpu() for n, p in net.named_parameters()}


def _compare(tag, hip, o32, o64=None):
    lp_h, g_h = hip
    lp_32, g_32 = o32
    r = {"log_probs_absmax": float(lp_32.abs().max()),
         "hip_vs_orc32_max": float((lp_h - lp_32).abs().max()),
         "hip_vs_orc32_rms": float((lp_h - lp_32).pow(2).mean().sqrt())}
    # besides the biases in front of a BatchNorm, a few BatchNorm biases have an exactly zero gradient too (the last BN of
    # an SA stack whose pooled output only ever feeds BatchNorm-ed layers: a constant shift of a channel is removed again):
    # their fp64 gradient is 1e-17 and a relative error means nothing
    gmax = max(float(v.norm()) for v in (o64[1] if o64 is not None else g_32).values())
    names = [n for n in g_32 if not _zero_grad_bias(n) and float((o64[1] if o64 is not None else g_32)[n].norm()) > 1e-9 * gmax]
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    r["grad_l2_hip_vs_orc32_worst"] = max(rel(g_h[n], g_32[n]) for n in names)
    r["grad_l2_hip_vs_orc32_median"] = float(np.median([rel(g_h[n], g_32[n]) for n in names]))
    r["grad_norm_hip_vs_orc32_worst"] = max(abs(float(g_h[n].norm() - g_32[n].norm())) / float(g_32[n].norm()) for n in names)
    if o64 is not None:
        lp_64, g_64 = o64
        r.update(hip_vs_fp64_max=float((lp_h - lp_64).abs().max()), orc32_vs_fp64_max=float((lp_32 - lp_64).abs().max()),
                 hip_vs_fp64_rms=float((lp_h - lp_64).pow(2).mean().sqrt()),
                 orc32_vs_fp64_rms=float((lp_32 - lp_64).pow(2).mean().sqrt()))
        eh = {n: rel(g_h[n], g_64[n]) for n in names}
        eo = {n: rel(g_32[n], g_64[n]) for n in names}
        r["grad_l2_hip_vs_fp64_worst"] = max(eh.values())
        r["grad_l2_orc32_vs_fp64_worst"] = max(eo.values())
        r["grad_l2_hip_vs_fp64_median"] = float(np.median(list(eh.values())))
        r["grad_l2_orc32_vs_fp64_median"] = float(np.median(list(eo.values())))
        r["grad_worst_ratio"] = max(eh[n] / max(eo[n], r["grad_l2_orc32_vs_fp64_median"]) for n in names)
        r["grad_worst_tensors"] = [(n, eh[n], eo[n], float(g_64[n].norm())) for n in sorted(names, key=lambda n: -eh[n])[:4]]
    _report(tag, r)
    return r


def _assert_yardstick(r):
    assert r["hip_vs_fp64_rms"] <= FACTOR * r["orc32_vs_fp64_rms"], r
    assert r["hip_vs_fp64_max"] <= FACTOR * r["orc32_vs_fp64_max"], r
    assert r["hip_vs_orc32_max"] <= REL_CAP * max(1.0, r["log_probs_absmax"]), r
    # gradients: every tensor as close to fp64 as the reference arithmetic's (a tensor the reference gets unusually
    # right is held to the median error instead), and the distribution as a whole
    assert r["grad_worst_ratio"] <= 1.5 * GRAD_FACTOR, r
    assert r["grad_l2_hip_vs_fp64_median"] <= GRAD_FACTOR * r["grad_l2_orc32_vs_fp64_median"], r
    assert r["grad_l2_hip_vs_fp64_worst"] <= GRAD_FACTOR * r["grad_l2_orc32_vs_fp64_worst"], r


@pytest.mark.parametrize("kind", ["ssg", "msg"])
def test_cfg3_full_size_network_vs_oracle_and_fp64(dev, kind):
    """BASELINE.json configs[2] at its real size (B=16 x 4096 x 9, train mode, dropout off), forward + loss + backward."""
    pts_np, lab_np = syn.kitti_batch(0, 16, 4096)
    pts, labels = torch.from_numpy(pts_np), torch.from_numpy(lab_np)
    net, orc = _nets(kind, dev)
    o32 = _run_oracle(orc, pts, labels, torch.float32)
    o64 = _run_oracle(orc, pts, labels, torch.float64)
    hip = _run_hip(net, pts, labels, dev)
    r = _compare("cfg3_%s_B16x4096" % kind, hip, o32, o64)
    _assert_yardstick(r)


@pytest.mark.parametrize("kind", ["ssg", "msg"])
def test_small_batch_network_vs_fp64_and_reference_self_noise(dev, kind):
    """The G6 size (B=2 x 1024): same yardstick, and the reference's own 8-vs-1-thread movement (g6_noise.npz, measured with
    the reference itself by tools/make_golden.py g6n) quoted next to the HIP path's distance to the reference's numbers."""
    g = golden("g6_nets.npz")
    noise = golden("g6_noise.npz")
    pts, labels = torch.from_numpy(g["points"]), torch.from_numpy(g["labels"])
    net, orc = _nets(kind, dev, seed=int(g["init_seed"]))
    o32 = _run_oracle(orc, pts, labels, torch.float32)
    o64 = _run_oracle(orc, pts, labels, torch.float64)
    hip = _run_hip(net, pts, labels, dev)
    r = _compare("g6_%s_B2x1024" % kind, hip, o32, o64)
    r["reference_8_vs_1_threads_max"] = float(noise[kind + "/log_probs_absdiff"])
    _report("g6_%s_B2x1024" % kind, r)
    assert r["hip_vs_fp64_rms"] <= FACTOR * r["orc32_vs_fp64_rms"], r
    assert r["hip_vs_fp64_max"] <= FACTOR * max(r["orc32_vs_fp64_max"], r["reference_8_vs_1_threads_max"]), r
    assert r["hip_vs_orc32_max"] <= FACTOR * r["reference_8_vs_1_threads_max"], r
    # gradients against the same fp64 evaluation (round 4: the sampled-tensor check of test_modules_gpu.py rests on this)
    assert r["grad_l2_hip_vs_fp64_median"] <= GRAD_FACTOR * r["grad_l2_orc32_vs_fp64_median"], r
    assert r["grad_l2_hip_vs_fp64_worst"] <= GRAD_FACTOR * r["grad_l2_orc32_vs_fp64_worst"], r
    assert r["grad_worst_ratio"] <= 1.5 * GRAD_FACTOR, r


def test_cfg5_single_cloud_vs_oracle(dev):
    """BASELINE.json configs[4], one 65 536-point cloud through both cfg5 networks (SSG with the reference's npoints:
    the FPS-over-64k stress; MSG with npoint x16: the ball-query / grouping stress), forward + backward vs the oracle."""
    pts_np, lab_np = syn.kitti_batch(7, 1, 65536)
    pts, labels = torch.from_numpy(pts_np), torch.from_numpy(lab_np)
    for kind, scale in (("ssg", 1), ("msg", 16)):
        net, orc = _nets(kind, dev, npoint_scale=scale)
        o32 = _run_oracle(orc, pts, labels, torch.float32)
        o64 = _run_oracle(orc, pts, labels, torch.float64)
        hip = _run_hip(net, pts, labels, dev)
        r = _compare("cfg5_%s_B1x65536" % kind, hip, o32, o64)
        assert r["hip_vs_orc32_max"] <= REL_CAP * max(1.0, r["log_probs_absmax"]), r
        assert r["hip_vs_fp64_rms"] <= FACTOR * r["orc32_vs_fp64_rms"], r
        assert r["hip_vs_fp64_max"] <= FACTOR * r["orc32_vs_fp64_max"], r
        assert r["grad_l2_hip_vs_fp64_median"] <= GRAD_FACTOR * r["grad_l2_orc32_vs_fp64_median"], r
        del net, orc, hip, o32, o64
        torch.cuda.empty_cache()


def test_cfg5_two_clouds_vs_oracle(dev):
    """BASELINE.json configs[4] with MORE than one cloud in the batch (round 2 verdict: the B=1 case never couples two dense
    scans): B=2 x 65 536 through the reference's SSG net -- two clouds share the cooperative multi-workgroup FPS launch, the
    Morton-ordered ball query and every BatchNorm's batch statistics -- forward + loss + backward against the oracle in fp32
    and in fp64 (same assertions as the single-cloud case)."""
    pts_np, lab_np = syn.kitti_batch(11, 2, 65536)
    pts, labels = torch.from_numpy(pts_np), torch.from_numpy(lab_np)
    net, orc = _nets("ssg", dev)
    o32 = _run_oracle(orc, pts, labels, torch.float32)
    o64 = _run_oracle(orc, pts, labels, torch.float64)
    hip = _run_hip(net, pts, labels, dev)
    r = _compare("cfg5_ssg_B2x65536", hip, o32, o64)
    assert r["hip_vs_orc32_max"] <= REL_CAP * max(1.0, r["log_probs_absmax"]), r
    assert r["hip_vs_fp64_rms"] <= FACTOR * r["orc32_vs_fp64_rms"], r
    assert r["hip_vs_fp64_max"] <= FACTOR * r["orc32_vs_fp64_max"], r
    assert r["grad_l2_hip_vs_fp64_median"] <= GRAD_FACTOR * r["grad_l2_orc32_vs_fp64_median"], r


@pytest.mark.parametrize("kind,scale", [("ssg", 1), ("msg", 16)])
def test_cfg5_full_batch_permutation_equivariance(dev, kind, scale):
    """BASELINE.json configs[4] as a WORKLOAD: B=8 x 65 536, forward + backward of the whole SA/FP stack (MSG: P up to
    8.4 M grouped rows per layer).  No CPU oracle finishes that in seconds; the size-independent property: permuting the
    clouds (with their FPS start indices) permutes outputs and input gradients and leaves the parameter gradients
    unchanged up to summation order."""
    torch.manual_seed(5)
    net = (M.PointNet2SemSeg(13, 6) if kind == "ssg" else M.PointNet2SemSegMsg(13, 6, npoint_scale=scale)).to(dev).train()
    pts = torch.from_numpy(syn.kitti_batch(100, 8, 65536)[0]).to(dev)
    g = torch.Generator().manual_seed(9)
    perm = torch.randperm(8, generator=g).to(dev)
    n_fps = [65536, 1024, 256, 64] if kind == "ssg" else [65536, 512 * scale]
    starts = [torch.randint(0, n, (8,), generator=g).to(dev) for n in n_fps]
    proj = (torch.randn(8, 128, 65536, generator=g) / 64).to(dev)

    class Feed:
        def __init__(self, seq):
            self.seq, self.i = seq, 0

        def take(self, B, N, device):
            self.i += 1
            assert int(self.seq[self.i - 1].max()) < N
            return self.seq[self.i - 1]

    def run(x, st, w):
        x = x.clone().requires_grad_(True)
        net.zero_grad(set_to_none=True)
        U.set_fps_start_feed(Feed(st))
        try:
            out = net.features(x)
        finally:
            U.set_fps_start_feed(None)
        (out * w).sum().backward()
        return out.detach(), x.grad.detach(), [p.grad.detach().clone() for p in net.parameters() if p.grad is not None]

    out_a, gin_a, gp_a = run(pts, starts, proj)
    out_b, gin_b, gp_b = run(pts[perm], [s[perm] for s in starts], proj[perm])
    assert out_a.shape == (8, 128, 65536) and bool(torch.isfinite(out_a).all())
    assert float((out_b - out_a[perm]).abs().max()) <= 5e-5 * max(1.0, float(out_a.abs().max()))
    d = (gin_b - gin_a[perm])[:, 3:]
    assert float(gin_a[:, 3:].norm()) > 0
    assert float(d.norm()) <= 1e-2 * float(gin_a[:, 3:].norm())
    typical = float(torch.stack([a.norm() for a in gp_a]).median())
    for a, b in zip(gp_a, gp_b):
        assert float((a - b).norm()) <= 2e-2 * float(a.norm()) + 1e-4 * typical, a.shape


def test_sample_and_group_values_vs_oracle(dev):
    """the composition fps -> gather -> ball query -> group -> centre -> cat([xyz_norm, feat]) against
    the oracle's primitives on the same start draw, values included (pointnet_util.py:110-137)."""
    pts, _ = syn.kitti_batch(9, 3, 2048)
    xyz_np = np.ascontiguousarray(pts[:, :3].transpose(0, 2, 1))
    feat_np = np.ascontiguousarray(pts[:, 3:].transpose(0, 2, 1))
    xyz, feat = torch.from_numpy(xyz_np).to(dev), torch.from_numpy(feat_np).to(dev)
    for with_feat in (True, False):
        torch.manual_seed(21)
        new_xyz, new_points, grouped_xyz, fps_idx = U.sample_and_group(128, 0.2, 32, xyz, feat if with_feat else None,
                                                                       returnfps=True)
        torch.manual_seed(21)
        start = T.draw_start(3, 2048).numpy()
        oi = G.farthest_point_sample(xyz_np, 128, start)
        o_new = G.index_points(xyz_np, oi)
        o_idx = G.query_ball_point(0.2, 32, xyz_np, o_new)
        o_rows = G.group(xyz_np, feat_np if with_feat else None, o_new, o_idx, True)
        assert (fps_idx.cpu().numpy() == oi).all()
        assert (new_xyz.cpu().numpy() == o_new).all()
        assert (grouped_xyz.cpu().numpy() == G.index_points(xyz_np, o_idx)).all()      # un-centred neighbours (:127)
        assert new_points.shape == o_rows.shape == (3, 128, 32, 9 if with_feat else 3)
        assert (new_points.cpu().numpy() == o_rows).all()                              # gather + exact fp32 subtraction
    # without returnfps: the two-tuple of :137
    torch.manual_seed(21)
    a, b = U.sample_and_group(128, 0.2, 32, xyz, feat)
    assert (a.cpu().numpy() == o_new).all() and b.shape == (3, 128, 32, 9)


ZOO = {
    "cls_msg": (lambda: M.PointNet2ClsMsg(), lambda: T.RefClsMsg(dropout=0.0), 1),
    "cls_ssg": (lambda: M.PointNet2ClsSsg(), lambda: T.RefClsSsg(dropout=0.0), 1),
    "partseg_ssg": (lambda: M.PointNet2PartSegSsg(50), lambda: T.RefPartSegSsg(50, dropout=0.0), 1),
    "partseg_msg": (lambda: M.PointNet2PartSegMsg_one_hot(50), lambda: T.RefPartSegMsgOneHot(50, dropout=0.0), 3),
}


@pytest.mark.parametrize("tag", sorted(ZOO))
def test_zoo_net_matches_reference_and_oracle(dev, tag):
    """The four other networks of model/pointnet2.py:7-139 (train mode, dropout off, B=2 x 1024): outputs against the
    REFERENCE's numbers (tests/golden/g10_zoo.npz), gradients against the oracle net on the same state."""
    g = golden("g10_zoo.npz")
    noise, g6n = golden("g10_noise.npz"), golden("g6_noise.npz")
    make, make_orc, n_in = ZOO[tag]
    torch.manual_seed(int(g["init_seed"]))
    orc = make_orc()
    net = make()
    net.load_state_dict(orc.state_dict())                     # identical keys: the reference's attribute names
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    net.to(dev).train()
    orc.train()
    pts = g["points"]
    xyz = torch.from_numpy(np.ascontiguousarray(pts[:, :3]))
    ins = [xyz, torch.from_numpy(np.ascontiguousarray(pts[:, 3:6])), torch.from_numpy(g["cls_label"])][:n_in]
    torch.manual_seed(int(g["fwd_seed"]))
    y = net(*[t.to(dev) for t in ins])
    ys = y if isinstance(y, tuple) else (y,)
    assert len(ys) == int(g[tag + "/n_out"])
    for i, t in enumerate(ys):
        assert tuple(t.shape) == tuple(g["%s/shape/%d" % (tag, i)])
        ref = g["%s/out/%d" % (tag, i)]
        mine = t.detach().cpu().numpy()
        mine = mine if mine.size <= 4096 else mine.reshape(-1)[::17]
        # B*N = 2048 behind up to six BatchNorm-coupled stages: the bound is TWICE what the reference moves against itself on
        # this very net when only its thread count changes (tests/golden/g10_noise.npz, tools/make_golden.py g10n: 7e-6 .. 9e-5),
        # and never below the smaller of the two G6 self-noise figures (nets of the same depth, g6_noise.npz)
        out_tol = max(2.0 * float(noise[tag + "/out_rel"]), float(min(g6n["ssg/log_probs_absdiff"], g6n["msg/log_probs_absdiff"])))
        assert np.abs(mine - ref).max() <= out_tol * max(1.0, np.abs(ref).max()), (tag, i, np.abs(mine - ref).max(), out_tol)
    gw = torch.randn(ys[0].shape, generator=torch.Generator().manual_seed(int(g["gw_seed"])))
    (ys[0] * gw.to(dev)).sum().backward()
    got = {n: p.grad for n, p in net.named_parameters()}
    # per-tensor gradient L2 norms: the reference's own 8-thread runs sit 1.5e-2 .. 1.7e-1 from its 1-thread run (a handful of
    # argmax / ReLU decisions of the tiny batch fall the other way): twice that, per net
    grad_tol = 2.0 * float(noise[tag + "/grad_l2_rel"])
    for n, l2 in zip(g[tag + "/grad_names"], g[tag + "/grad_l2"]):
        n = str(n)
        if _zero_grad_bias(n):
            continue
        mine = float(got[n].double().norm())
        assert abs(mine - l2) <= grad_tol * l2 + 1e-6 * float(g[tag + "/grad_l2"].max()), (n, mine, l2, grad_tol)


def test_partseg_msg_shape_of_the_reference_self_test(dev):
    """The reference's only in-repo pin (model/pointnet2.py:179-187): input (8,3,2048), label (8,16) ->
    torch.Size([8, 2048, 50])."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(8, 3, 2048, generator=g).to(dev)
    label = torch.randn(8, 16, generator=g).to(dev)
    net = M.PointNet2PartSegMsg_one_hot(num_classes=50).to(dev)
    out = net(x, x, label)
    assert out.size() == torch.Size([8, 2048, 50])
    assert bool(torch.isfinite(out).all())
    assert float((out.exp().sum(-1) - 1).abs().max()) <= 1e-4          # rows are log-probabilities
