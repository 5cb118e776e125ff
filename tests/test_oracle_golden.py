"""CPU: the oracle (oracle/) against the committed golden vectors produced by the reference."""
import hashlib

import numpy as np
import pytest
import torch

from conftest import golden
from oracle import geometry as G
from oracle import torch_ref as T


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_fps_golden():
    g = golden("g1_fps.npz")
    for tag in g["cases"]:
        xyz, start, ref = g[tag + "/xyz"], g[tag + "/start"], g[tag + "/idx"]
        mine = G.farthest_point_sample(xyz, ref.shape[1], start)
        assert (mine == ref).all(), tag
    # exhausted clouds return index 0 forever (only 20 distinct points in 'dups256')
    assert (g["dups256/idx"][:, 25:] == 0).all()


def test_ball_query_golden():
    g = golden("g2_ball.npz")
    for tag in g["cases"]:
        fam, rest = tag.split("/")
        r, k = rest[1:].split("_k")
        mine = G.query_ball_point(float(r), int(k), g[fam + "/xyz"], g[fam + "/new_xyz"])
        assert (mine == g[tag]).all(), tag
    for r in (0.1, 0.2, 0.4, 0.8):
        pre = "edge/r%g/" % r
        mine = G.query_ball_point(r, 4, g[pre + "xyz"], g[pre + "new_xyz"])
        assert (mine == g[pre + "idx"]).all()
        assert list(mine[0, 0]) == [1, 2, 1, 1]        # d == r^2 inside, +1 ulp outside


def test_ball_query_empty_and_nsample():
    xyz = np.random.default_rng(0).uniform(-1, 1, (1, 50, 3)).astype(np.float32)
    far = np.full((1, 1, 3), 9.0, np.float32)
    assert (G.query_ball_point(0.1, 8, xyz, far) == 50).all()
    with pytest.raises(RuntimeError):
        G.query_ball_point(0.1, 51, xyz, far)
    with pytest.raises(IndexError):
        G.index_points(xyz, np.full((1, 2), 50))


def test_square_distance_bits():
    g = golden("g3_sqdist.npz")
    d = G.square_distance(g["new_xyz"], g["xyz"])
    assert hashlib.sha256(bits(d).tobytes()).hexdigest() == str(g["sha256"])
    assert (bits(d)[0, ::16, :] == g["sample_bits"]).all()
    assert d.min() < 0          # the expanded form goes slightly negative at coincident points


def test_three_nn_interp_golden():
    g = golden("g4_interp.npz")
    for tag in "abc":
        idx, dist = G.three_nn(g[tag + "/xyz1"], g[tag + "/xyz2"])
        assert (bits(dist) == bits(g[tag + "/dist3"])).all()
        out = G.three_interpolate(g[tag + "/points2"], idx, G.three_weights(dist))
        assert np.abs(out - g[tag + "/interp"]).max() <= 2e-6


MODULE_CASES = {
    "sa": lambda: T.RefSetAbstraction(256, 0.2, 32, 9, [32, 32, 64], False),
    "sa_nofeat": lambda: T.RefSetAbstraction(128, 0.4, 16, 3, [16, 32], False),
    "sa_all": lambda: T.RefSetAbstraction(None, None, None, 9, [32, 64], True),
    "msg": lambda: T.RefSetAbstractionMsg(128, [0.1, 0.2, 0.4], [16, 32, 64], 6, [[16, 32], [32, 48], [32, 196]]),
    "fp": lambda: T.RefFeaturePropagation(30, [32, 16]),
    "fp_noskip": lambda: T.RefFeaturePropagation(24, [32, 32, 16]),
    "fp_s1": lambda: T.RefFeaturePropagation(30, [16]),
}
MODULE_INPUTS = {"sa": ["xyz", "points"], "sa_nofeat": ["xyz", "points"], "sa_all": ["xyz", "points"],
                 "msg": ["xyz", "points"], "fp": ["xyz1", "xyz2", "points1", "points2"],
                 "fp_noskip": ["xyz1", "xyz2", "points1", "points2"], "fp_s1": ["xyz1", "xyz2", "points1", "points2"]}


def module_case(g, tag):
    """-> (state0 dict, inputs list (None where absent), seed)."""
    state = {k[len(tag) + 8:]: g[k] for k in g.files if k.startswith(tag + "/state0/")}
    ins = [g[tag + "/in/" + n] if (tag + "/in/" + n) in g.files else None for n in MODULE_INPUTS[tag]]
    return state, ins, int(g[tag + "/seed"])


@pytest.mark.parametrize("tag", sorted(MODULE_CASES))
def test_oracle_modules_golden(tag):
    g = golden("g5_modules.npz")
    state, ins, seed = module_case(g, tag)
    mod = MODULE_CASES[tag]()
    T.load_numpy_state(mod, state)
    mod.train()
    names = MODULE_INPUTS[tag]
    tens = [None if a is None else torch.from_numpy(a).requires_grad_(("gin/" + n) in "".join(
        k for k in g.files if k.startswith(tag + "/gin/"))) for n, a in zip(names, ins)]
    torch.manual_seed(seed)
    y = mod(*tens)
    ys = y if isinstance(y, tuple) else (y,)
    for i, t in enumerate(ys):
        assert np.abs(t.detach().numpy() - g["%s/out/%d" % (tag, i)]).max() <= 5e-6
    (ys[-1] * torch.from_numpy(g[tag + "/gw"])).sum().backward()
    for n, t in zip(names, tens):
        key = "%s/gin/%s" % (tag, n)
        if key in g.files:
            ref = g[key]
            assert np.abs(t.grad.numpy() - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1e-12)
    for k, p in mod.named_parameters():
        if "conv" in k and k.endswith("bias"):
            continue
        ref = g["%s/gpar/%s" % (tag, k)]
        assert np.abs(p.grad.numpy() - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1e-12), k
    for k, v in mod.state_dict().items():
        key = "%s/state1/%s" % (tag, k)
        if key in g.files:
            assert np.allclose(v.numpy(), g[key], rtol=1e-5, atol=1e-6), k


def test_rng_draw_order():
    g = golden("g6_nets.npz")
    torch.manual_seed(int(g["fwd_seed"]))
    starts = np.stack([T.draw_start(2, n).numpy() for n in (1024, 1024, 256, 64)])
    assert (starts == g["ssg/starts"]).all()


def test_loader_restatement_golden():
    """oracle.train_ref.prepare_cloud against the reference's pcd_normalize / pcd_jitter / resampling outputs."""
    from oracle import train_ref as TR
    g = golden("g8_train.npz")
    for tag in g["loader_cases"]:
        np.random.seed(int(g[tag + "/np_seed"]))
        pts, lab, _, choice = TR.prepare_cloud(g[tag + "/raw"], g[tag + "/label"], g[tag + "/points"].shape[0],
                                               bool(g[tag + "/train"]))
        assert (choice == g[tag + "/choice"]).all(), tag
        assert (bits(pts) == bits(g[tag + "/points"])).all(), tag
        assert (lab == g[tag + "/labels"]).all(), tag
    # out-of-range rows are clipped to [-1, 1] before the jitter is added (SemKITTI_Loader.py:29)
    n = TR.normalize(g["eval/raw"][:5])
    assert n[0].tolist() == [1.0, -1.0, 1.0, 1.0] and n[1].tolist() == [-1.0, 1.0, -1.0, -1.0]


def test_adam_restatement_golden():
    """oracle.train_ref.adam_step against 12 steps of torch.optim.Adam as semseg.py:106-111 configures it."""
    from oracle import train_ref as TR
    g = golden("g8_train.npz")
    p = g["adam/param0"].copy()
    m, v = np.zeros_like(p), np.zeros_like(p)
    for t, (grad, lr, ref) in enumerate(zip(g["adam/grads"], g["adam/lr"], g["adam/after"]), 1):
        TR.adam_step(p, grad, m, v, t, lr=float(lr), weight_decay=float(g["adam/weight_decay"]))
        assert np.abs(p - ref).max() <= 2e-7 * np.abs(ref).max(), t
    assert np.abs(m - g["adam/exp_avg"]).max() <= 1e-6 * np.abs(g["adam/exp_avg"]).max()
    assert np.abs(v - g["adam/exp_avg_sq"]).max() <= 1e-6 * np.abs(g["adam/exp_avg_sq"]).max()
    # and against torch.optim.Adam itself, which travels with the image (no weight decay, another lr)
    torch.manual_seed(3)
    q = torch.nn.Parameter(torch.randn(1001))
    opt = torch.optim.Adam([q], lr=3e-3, betas=(0.9, 0.999), eps=1e-08)
    p = q.detach().numpy().copy()
    m, v = np.zeros_like(p), np.zeros_like(p)
    for t in range(1, 6):
        q.grad = torch.randn(1001)
        opt.step()
        TR.adam_step(p, q.grad.numpy(), m, v, t, lr=3e-3)
        assert np.abs(p - q.detach().numpy()).max() <= 2e-7 * np.abs(p).max()


def test_aten_geometry_equals_c_restatement():
    """oracle/aten_geometry.py (the reference's operator sequence, timed as the CPU baseline) returns the C
    restatement's indices bit for bit, and the oracle modules give the same outputs in either mode."""
    from oracle.aten_geometry import AtenGeometry as A
    from pointnet12_amd import synthetic as syn
    pts, _ = syn.kitti_batch(33, 2, 1024)
    xyz = np.ascontiguousarray(pts[:, :3].transpose(0, 2, 1))
    t = torch.from_numpy(xyz)
    start = np.array([3, 700])
    f = A.fps(t, 128, torch.from_numpy(start))
    assert (f.numpy() == G.farthest_point_sample(xyz, 128, start)).all()
    new = A.gather(t, f)
    assert (new.numpy() == G.index_points(xyz, f.numpy())).all()
    for r, k in ((0.1, 16), (0.4, 64)):
        assert (A.ball(r, k, t, new).numpy() == G.query_ball_point(r, k, xyz, new.numpy())).all()
    assert (bits(A.pair_sqdist(new, t).numpy()) == bits(G.square_distance(new.numpy(), xyz))).all()
    p2 = torch.randn(2, 128, 5, generator=torch.Generator().manual_seed(0))
    interp, _ = A.three_nn_interp(t, new, p2)
    oi, od = G.three_nn(xyz, new.numpy())
    assert np.abs(interp.numpy() - G.three_interpolate(p2.numpy(), oi, G.three_weights(od))).max() <= 2e-6
    outs = []
    for mode in ("c", "aten"):
        T.set_geometry(mode)
        try:
            torch.manual_seed(0)
            net = T.RefSSGSemSeg(13, 6, dropout=0.0).train()
            torch.manual_seed(1)
            outs.append(net(torch.from_numpy(pts)).detach())
        finally:
            T.set_geometry("c")
    assert float((outs[0] - outs[1]).abs().max()) <= 1e-5


def test_shipped_checkpoint_eval_golden():
    """G7: the reference's shipped KITTI checkpoint in eval mode.  The fixture holds the input cloud, the
    REFERENCE's log-probabilities and the checkpoint's sha256; the weights themselves never enter the repo, so this runs only
    where the reference tree is mounted (the development container) and is skipped on the GPU box.  It pins, against the
    reference's own numbers: the 156 `module.`-prefixed keys loading into the oracle net AND into this package's net class
    (the state_dict contract of SURVEY.md 8(b)), and the oracle's eval-mode arithmetic (running statistics, no dropout)."""
    import os
    path = os.path.join(os.environ.get("PN2_REFERENCE", "/root/reference"), "checkpoints", "pointnet2-inview-0.55884-0001.pth")
    if not os.path.exists(path):
        pytest.skip("reference checkpoint not mounted (GPU box): the fixture's input/output pair needs its weights")
    g = golden("g7_checkpoint_eval.npz")
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == str(g["sha256"])
    sd = torch.load(path, map_location="cpu")
    assert len(sd) == int(g["n_keys"]) == 156 and all(k.startswith("module.") for k in sd)
    sd = {k[len("module."):]: v for k, v in sd.items()}
    orc = T.RefSSGSemSeg(19, 1)
    orc.load_state_dict(sd)                      # strict: every key and shape of the reference's SA / FP / head modules
    orc.eval()
    from pointnet12_amd import pointnet2 as M    # the product's model zoo: same attribute names, same shapes (no GPU needed)
    net = M.PointNet2SemSeg(19, 1)
    assert net.load_state_dict(sd, strict=True) is not None
    with torch.no_grad():
        torch.manual_seed(int(g["seed"]))
        mine = orc(torch.from_numpy(g["points"])).numpy()
    assert mine.shape == g["log_probs"].shape == (1, 2048, 19)
    assert np.abs(mine - g["log_probs"]).max() <= 2e-5
