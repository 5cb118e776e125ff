#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE on CPU (development container only).

Imports /root/reference/model/pointnet_util.py + pointnet2.py unmodified, feeds them the
synthetic inputs of pointnet12_amd.synthetic, and stores inputs + reference outputs.  While
doing so it pins the oracle: every index tensor and the raw fp32 distance matrix produced by
oracle/ must be bit-equal to the reference's, module outputs within 5e-6 (the reference's own
8-thread-vs-1-thread noise), or this script aborts.  Nothing from the reference is written to
the repo except numbers it computed.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PN2_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

from model import pointnet_util as R          # noqa: E402  (the reference)
from model import pointnet2 as R2             # noqa: E402
from oracle import geometry as G              # noqa: E402
from oracle import torch_ref as T             # noqa: E402
from pointnet12_amd import synthetic as syn   # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def check(cond, what):
    if not cond:
        raise SystemExit("ORACLE MISMATCH: " + what)
    print("  ok:", what)


def xyz_of(points_cf):
    """[B,C,N] -> the permuted [B,N,3] view the modules hand to the primitives."""
    return torch.from_numpy(points_cf[:, :3, :]).permute(0, 2, 1)


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.1f KB)" % (name, os.path.getsize(path) / 1024))


# ----------------------------------------------------------------------------- G1 FPS
def g1_fps():
    print("G1 farthest_point_sample")
    out = {}
    cases = []
    for tag, B, N, S, kind in [("kitti4096", 2, 4096, 1024, "kitti"), ("kitti1024", 2, 1024, 256, "kitti"),
                               ("kitti64", 4, 64, 16, "kitti"), ("unif1000", 2, 1000, 250, "uniform"),
                               ("dups256", 2, 256, 64, "dups"), ("over64", 1, 64, 100, "kitti"),
                               ("kitti8192", 1, 8192, 64, "kitti")]:
        if kind == "kitti":
            xyz = xyz_of(syn.kitti_batch(40, B, N)[0]).contiguous()
        elif kind == "uniform":
            xyz = xyz_of(syn.uniform_batch(11, B, N)[0]).contiguous()
        else:  # only 20 distinct points: FPS exhausts them and then returns index 0 forever
            base = xyz_of(syn.uniform_batch(12, B, 20)[0])
            pick = torch.from_numpy(np.random.default_rng(5).integers(0, 20, size=(B, N)))
            xyz = torch.stack([base[b][pick[b]] for b in range(B)]).contiguous()
        torch.manual_seed(100 + N)
        state = torch.get_rng_state()
        ref = R.farthest_point_sample(xyz, S).numpy()
        torch.set_rng_state(state)
        start = T.draw_start(B, N).numpy()
        check((ref[:, 0] == start).all(), "%s: start draw reproduces reference column 0" % tag)
        mine = G.farthest_point_sample(xyz.numpy(), S, start)
        check((mine == ref).all(), "%s: FPS indices bit-equal" % tag)
        out[tag + "/xyz"] = xyz.numpy()
        out[tag + "/start"] = start
        out[tag + "/idx"] = ref.astype(np.int32)
        cases.append(tag)
    out["cases"] = np.array(cases)
    save("g1_fps.npz", **out)


# ----------------------------------------------------------------------------- G2 ball query
def boundary_case(radius, rng):
    """Points whose fp32 distance to the centre is exactly r^2, and one ulp either side.

    The centre is the origin so the expanded form collapses to n(p) = ((x*x + y*y) + z*z)
    and every fp32 value near r^2 is reachable; x is walked over the floats around r and a
    small y nudges the sum.  (Statistical boundary coverage at general centres comes from the
    KITTI cases: about one pair per 8 M sits within an ulp of r^2.)
    """
    r2 = np.float32(radius ** 2)
    want = {"eq": r2, "lo": np.nextafter(r2, np.float32(0)), "hi": np.nextafter(r2, np.float32(1))}
    centre = np.zeros((1, 3), np.float32)
    x0 = np.float32(radius)
    xs = [x0]
    for _ in range(64):
        xs.append(np.nextafter(xs[-1], np.float32(0)))
    xs = np.array(xs, np.float32)
    ys = np.concatenate([[0.0], np.float32(radius) * 2.0 ** -np.arange(8, 14, 0.25)]).astype(np.float32)
    cand = np.array([[x, y, 0.0] for x in xs for y in ys], np.float32)
    d = G.square_distance(centre[None], cand[None])[0, 0]
    found = {}
    for k, v in want.items():
        hit = np.nonzero(d == v)[0]
        if not hit.size:
            raise SystemExit("could not construct boundary point %s for r=%g" % (k, radius))
        found[k] = cand[hit[0]]
    filler = np.float32(radius * 3) + rng.uniform(0, 0.01, size=(13, 3)).astype(np.float32)
    xyz = np.concatenate([found["hi"][None], found["eq"][None], found["lo"][None], filler]).astype(np.float32)
    return xyz[None], centre[None]          # [1,16,3], [1,1,3]


def g2_ball():
    print("G2 query_ball_point")
    out = {}
    cases = []
    pts = syn.kitti_batch(60, 2, 4096)[0]
    xyz = xyz_of(pts)
    torch.manual_seed(3)
    new_xyz = R.index_points(xyz, R.farthest_point_sample(xyz, 512))
    out["kitti/xyz"] = xyz.contiguous().numpy()
    out["kitti/new_xyz"] = new_xyz.numpy()
    for radius, K in [(0.1, 16), (0.1, 32), (0.2, 32), (0.2, 64), (0.4, 64), (0.4, 128), (0.8, 128), (0.05, 32)]:
        ref = R.query_ball_point(radius, K, xyz, new_xyz).numpy()
        mine = G.query_ball_point(radius, K, xyz.numpy(), new_xyz.numpy())
        tag = "kitti/r%g_k%d" % (radius, K)
        check((mine == ref).all(), tag + ": ball-query indices bit-equal")
        out[tag] = ref.astype(np.int32)
        cases.append(tag)
    # sparse uniform cloud: almost every group is padding
    upts = syn.uniform_batch(21, 2, 1024)[0]
    uxyz = xyz_of(upts)
    torch.manual_seed(4)
    unew = R.index_points(uxyz, R.farthest_point_sample(uxyz, 128))
    out["unif/xyz"] = uxyz.contiguous().numpy()
    out["unif/new_xyz"] = unew.numpy()
    for radius, K in [(0.1, 32), (0.4, 64)]:
        ref = R.query_ball_point(radius, K, uxyz, unew).numpy()
        mine = G.query_ball_point(radius, K, uxyz.numpy(), unew.numpy())
        tag = "unif/r%g_k%d" % (radius, K)
        check((mine == ref).all(), tag + ": ball-query indices bit-equal")
        out[tag] = ref.astype(np.int32)
        cases.append(tag)
    # d == r^2 is inside, one ulp above is outside
    rng = np.random.default_rng(77)
    for radius in (0.1, 0.2, 0.4, 0.8):
        bx, bc = boundary_case(radius, rng)
        ref = R.query_ball_point(radius, 4, torch.from_numpy(bx), torch.from_numpy(bc)).numpy()
        mine = G.query_ball_point(radius, 4, bx, bc)
        check((ref[0, 0] == np.array([1, 2, 1, 1])).all(), "boundary r=%g: reference keeps d==r^2, drops +1ulp" % radius)
        check((mine == ref).all(), "boundary r=%g: oracle agrees" % radius)
        out["edge/r%g/xyz" % radius] = bx
        out["edge/r%g/new_xyz" % radius] = bc
        out["edge/r%g/idx" % radius] = ref.astype(np.int32)
    # empty ball -> N everywhere (and the reference then raises in index_points)
    far = torch.full((1, 1, 3), 5.0)
    ref = R.query_ball_point(0.1, 8, uxyz[:1], far).numpy()
    mine = G.query_ball_point(0.1, 8, uxyz[:1].numpy(), far.numpy())
    check((ref == 1024).all() and (mine == ref).all(), "empty ball gives N in every slot")
    # the double-vs-float threshold compare is identical for every radius in the model zoo
    for radius in (0.05, 0.1, 0.2, 0.4, 0.8):
        d = R.square_distance(new_xyz, xyz)
        check(bool(((d > radius ** 2) == (d > torch.tensor(np.float32(radius ** 2)))).all()),
              "r=%g: fp32 threshold compare equals the reference's double compare" % radius)
    out["cases"] = np.array(cases)
    save("g2_ball.npz", **out)


# ----------------------------------------------------------------------------- G3 distance bits
def g3_sqdist():
    print("G3 square_distance bit pattern")
    pts = syn.kitti_batch(80, 1, 4096)[0]
    xyz = xyz_of(pts)
    torch.manual_seed(9)
    new_xyz = R.index_points(xyz, R.farthest_point_sample(xyz, 1024))
    ref = R.square_distance(new_xyz, xyz).numpy()
    mine = G.square_distance(new_xyz.numpy(), xyz.numpy())
    check((bits(ref) == bits(mine)).all(), "1024x4096 distance matrix bit-equal")
    rev = R.square_distance(xyz, new_xyz).numpy()
    check((bits(rev) == bits(G.square_distance(xyz.numpy(), new_xyz.numpy()))).all(), "4096x1024 (FP direction) bit-equal")
    save("g3_sqdist.npz", xyz=xyz.contiguous().numpy(), new_xyz=new_xyz.numpy(),
         sha256=np.array(hashlib.sha256(bits(ref).tobytes()).hexdigest()),
         rows=np.arange(0, 1024, 16), sample_bits=bits(ref)[0, ::16, :])


# ----------------------------------------------------------------------------- G4 three-NN interpolation
def g4_interp():
    print("G4 three-NN + inverse-distance interpolation")
    out = {}
    for tag, N, S, D in [("a", 1024, 256, 16), ("b", 256, 64, 8), ("c", 64, 3, 5)]:
        pts = syn.kitti_batch(90, 2, N)[0]
        xyz1 = xyz_of(pts)
        torch.manual_seed(17)
        xyz2 = R.index_points(xyz1, R.farthest_point_sample(xyz1, S))      # xyz2 is a subset of xyz1
        points2 = torch.randn(2, S, D, generator=torch.Generator().manual_seed(5))
        dists, idx = R.square_distance(xyz1, xyz2).sort(dim=-1)
        dists, idx = dists[:, :, :3].clone(), idx[:, :, :3]
        raw = dists.clone().numpy()
        dists[dists < 1e-10] = 1e-10
        w = 1.0 / dists
        w = w / torch.sum(w, dim=-1).view(2, N, 1)
        interp = torch.sum(R.index_points(points2, idx) * w.view(2, N, 3, 1), dim=2).numpy()
        oi, od = G.three_nn(xyz1.numpy(), xyz2.numpy())
        check((bits(od) == bits(raw)).all(), tag + ": top-3 distances bit-equal")
        ow = G.three_weights(od)
        check(np.abs(ow - w.numpy()).max() <= 1.2e-7, tag + ": weights within 1 ulp")
        mine = G.three_interpolate(points2.numpy(), oi, ow)
        check(np.abs(mine - interp).max() <= 2e-6, tag + ": interpolated features within 2e-6 (%.2e)" % np.abs(mine - interp).max())
        out.update({tag + "/xyz1": xyz1.contiguous().numpy(), tag + "/xyz2": xyz2.numpy(),
                    tag + "/points2": points2.numpy(), tag + "/dist3": raw, tag + "/interp": interp})
    save("g4_interp.npz", **out)


# ----------------------------------------------------------------------------- G5 modules
def _grads(module):
    return {k: p.grad.detach().numpy().copy() for k, p in module.named_parameters()}


def _module_case(tag, ref, orc, inputs, grad_names, seed, out):
    """Run reference and oracle module fwd+bwd on the same inputs; store the reference's numbers."""
    orc.load_state_dict(ref.state_dict())
    state0 = T.numpy_state(ref)
    res = {}
    for which, mod in (("ref", ref), ("orc", orc)):
        mod.train()
        ins = [None if t is None else t.clone().requires_grad_(n in grad_names) for n, t in inputs]
        torch.manual_seed(seed)
        y = mod(*ins)
        ys = y if isinstance(y, tuple) else (y,)
        feat = ys[-1]
        gw = torch.randn(feat.shape, generator=torch.Generator().manual_seed(seed + 1))
        (feat * gw).sum().backward()
        res[which] = dict(outs=[t.detach().numpy().copy() for t in ys], gw=gw.numpy(),
                          gin={n: t.grad.numpy().copy() for (n, _), t in zip(inputs, ins) if t is not None and t.grad is not None},
                          gpar=_grads(mod), state1=T.numpy_state(mod))
    r, o = res["ref"], res["orc"]
    for a, b in zip(r["outs"], o["outs"]):
        check(np.abs(a - b).max() <= 5e-6, "%s: oracle module output within 5e-6 (%.2e)" % (tag, np.abs(a - b).max()))
    for n in r["gin"]:
        e = np.abs(r["gin"][n] - o["gin"][n]).max() / max(np.abs(r["gin"][n]).max(), 1e-12)
        check(e <= 2e-5, "%s: d/d%s within 2e-5 of max (%.2e)" % (tag, n, e))
    for n in r["gpar"]:
        if "conv" in n and n.endswith("bias"):
            continue          # mathematically zero under training-mode BN: pure rounding noise
        e = np.abs(r["gpar"][n] - o["gpar"][n]).max() / max(np.abs(r["gpar"][n]).max(), 1e-12)
        check(e <= 2e-5, "%s: grad %s within 2e-5 of max (%.2e)" % (tag, n, e))
    for k in r["state1"]:
        check(np.allclose(r["state1"][k], o["state1"][k], rtol=1e-5, atol=1e-6), "%s: buffer %s after step" % (tag, k))
    for k, v in state0.items():
        out["%s/state0/%s" % (tag, k)] = v
    for k, v in r["state1"].items():
        if "running" in k or "tracked" in k:
            out["%s/state1/%s" % (tag, k)] = v
    for n, t in inputs:
        if t is not None:
            out["%s/in/%s" % (tag, n)] = t.numpy()
    for i, a in enumerate(r["outs"]):
        out["%s/out/%d" % (tag, i)] = a
    out["%s/gw" % tag] = r["gw"]
    for n, a in r["gin"].items():
        out["%s/gin/%s" % (tag, n)] = a
    for n, a in r["gpar"].items():
        out["%s/gpar/%s" % (tag, n)] = a
    out["%s/seed" % tag] = np.int64(seed)


def g5_modules():
    print("G5 modules")
    out = {}
    pts = torch.from_numpy(syn.kitti_batch(120, 2, 1024)[0])
    xyz, feat = pts[:, :3, :], pts[:, 3:, :]

    torch.manual_seed(1)
    _module_case("sa", R.PointNetSetAbstraction(256, 0.2, 32, 9, [32, 32, 64], False),
                 T.RefSetAbstraction(256, 0.2, 32, 9, [32, 32, 64], False),
                 [("xyz", xyz), ("points", feat)], {"points"}, 11, out)
    torch.manual_seed(2)
    _module_case("sa_nofeat", R.PointNetSetAbstraction(128, 0.4, 16, 3, [16, 32], False),
                 T.RefSetAbstraction(128, 0.4, 16, 3, [16, 32], False),
                 [("xyz", xyz), ("points", None)], set(), 12, out)
    torch.manual_seed(3)
    _module_case("sa_all", R.PointNetSetAbstraction(None, None, None, 9, [32, 64], True),
                 T.RefSetAbstraction(None, None, None, 9, [32, 64], True),
                 [("xyz", xyz[:, :, :200].contiguous()), ("points", feat[:, :, :200].contiguous())], {"points"}, 13, out)
    torch.manual_seed(4)
    _module_case("msg", R.PointNetSetAbstractionMsg(128, [0.1, 0.2, 0.4], [16, 32, 64], 6, [[16, 32], [32, 48], [32, 196]]),
                 T.RefSetAbstractionMsg(128, [0.1, 0.2, 0.4], [16, 32, 64], 6, [[16, 32], [32, 48], [32, 196]]),
                 [("xyz", xyz), ("points", feat)], {"points"}, 14, out)
    # FP: xyz2 is the FPS subset of xyz1 (as in every network), with and without skip features
    torch.manual_seed(21)
    sub = R.farthest_point_sample(xyz.permute(0, 2, 1), 128)
    xyz2 = R.index_points(xyz.permute(0, 2, 1), sub).permute(0, 2, 1).contiguous()
    points2 = torch.randn(2, 24, 128, generator=torch.Generator().manual_seed(8))
    torch.manual_seed(5)
    _module_case("fp", R.PointNetFeaturePropagation(6 + 24, [32, 16]), T.RefFeaturePropagation(6 + 24, [32, 16]),
                 [("xyz1", xyz), ("xyz2", xyz2), ("points1", feat), ("points2", points2)], {"points1", "points2"}, 15, out)
    torch.manual_seed(6)
    _module_case("fp_noskip", R.PointNetFeaturePropagation(24, [32, 32, 16]), T.RefFeaturePropagation(24, [32, 32, 16]),
                 [("xyz1", xyz), ("xyz2", xyz2), ("points1", None), ("points2", points2)], {"points2"}, 16, out)
    torch.manual_seed(7)
    _module_case("fp_s1", R.PointNetFeaturePropagation(6 + 24, [16]), T.RefFeaturePropagation(6 + 24, [16]),
                 [("xyz1", xyz[:, :, :128].contiguous()), ("xyz2", torch.zeros(2, 3, 1)),
                  ("points1", feat[:, :, :128].contiguous()), ("points2", points2[:, :, :1].contiguous())],
                 {"points1", "points2"}, 17, out)
    save("g5_modules.npz", **out)


# ----------------------------------------------------------------------------- G6 networks, G8 rng order
class RefMSGSemSegFromReference(torch.nn.Module):
    """MSG-SemSeg of SURVEY.md §8(d), composed from REFERENCE modules (only ever lives here)."""

    def __init__(self, num_classes, d):
        super().__init__()
        self.sa1 = R.PointNetSetAbstractionMsg(512, [0.1, 0.2, 0.4], [32, 64, 128], d, [[32, 32, 64], [64, 64, 128], [64, 96, 128]])
        self.sa2 = R.PointNetSetAbstractionMsg(128, [0.4, 0.8], [64, 128], 320, [[128, 128, 256], [128, 196, 256]])
        self.sa3 = R.PointNetSetAbstraction(None, None, None, 515, [256, 512, 1024], True)
        self.fp3 = R.PointNetFeaturePropagation(1536, [256, 256])
        self.fp2 = R.PointNetFeaturePropagation(576, [256, 128])
        self.fp1 = R.PointNetFeaturePropagation(131 + d, [128, 128])
        self.conv1 = torch.nn.Conv1d(128, 128, 1)
        self.bn1 = torch.nn.BatchNorm1d(128)
        self.drop1 = torch.nn.Dropout(0.5)
        self.conv2 = torch.nn.Conv1d(128, num_classes, 1)

    def forward(self, points):
        xyz, feat = points[:, :3, :], points[:, 3:, :]
        x1, f1 = self.sa1(xyz, feat)
        x2, f2 = self.sa2(x1, f1)
        x3, f3 = self.sa3(x2, f2)
        f2 = self.fp3(x2, x3, f2, f3)
        f1 = self.fp2(x1, x2, f1, f2)
        f0 = self.fp1(xyz, x1, torch.cat([xyz, feat], 1), f1)
        x = self.drop1(torch.relu(self.bn1(self.conv1(f0))))
        return torch.log_softmax(self.conv2(x), dim=1).permute(0, 2, 1)


def state_digest(module):
    h = hashlib.sha256()
    for k, v in module.state_dict().items():
        h.update(k.encode())
        h.update(v.detach().numpy().tobytes())
    return h.hexdigest()


def g6_nets():
    print("G6 networks / G8 rng draw order")
    out = {}
    pts_np, labels = syn.kitti_batch(200, 2, 1024)
    pts = torch.from_numpy(pts_np)
    for tag, make_ref, make_orc in [("ssg", lambda: R2.PointNet2SemSeg(13, 6), lambda: T.RefSSGSemSeg(13, 6, dropout=0.0)),
                                    ("msg", lambda: RefMSGSemSegFromReference(13, 6), lambda: T.RefMSGSemSeg(13, 6, dropout=0.0))]:
        torch.manual_seed(1234)
        ref = make_ref()
        ref.drop1.p = 0.0
        torch.manual_seed(1234)
        orc = make_orc()
        check(state_digest(ref) == state_digest(orc), tag + ": seeded init of oracle net equals reference net")
        check(T.count_params(ref) == {"ssg": 968173, "msg": 1735001}[tag], tag + ": parameter count")
        init_digest = state_digest(ref)          # before any forward pass touches the BN buffers
        res = {}
        for which, net in (("ref", ref), ("orc", orc)):
            net.train()
            torch.manual_seed(4321)
            lp = net(pts)
            loss = T.seg_loss(lp, labels)
            loss.backward()
            res[which] = (lp.detach().numpy(), float(loss), {k: p.grad.numpy().copy() for k, p in net.named_parameters()})
        d = np.abs(res["ref"][0] - res["orc"][0]).max()
        check(d <= 2e-5, "%s: oracle net log-probs within 2e-5 of reference (%.2e)" % (tag, d))
        check(abs(res["ref"][1] - res["orc"][1]) <= 1e-6, "%s: loss" % tag)
        out[tag + "/init_sha256"] = np.array(init_digest)
        out[tag + "/log_probs"] = res["ref"][0]
        out[tag + "/loss"] = np.float64(res["ref"][1])
        names = sorted(res["ref"][2])
        out[tag + "/grad_names"] = np.array(names)
        out[tag + "/grad_l2"] = np.array([np.linalg.norm(res["ref"][2][n].astype(np.float64)) for n in names])
        out[tag + "/grad_absmax"] = np.array([np.abs(res["ref"][2][n]).max() for n in names])
        # keep a few full gradient tensors (first and last stage) for a sharper check
        for n in names:
            if n.startswith(("sa1.", "conv2.", "fp1.")) and n.endswith("weight"):
                out[tag + "/grad/" + n] = res["ref"][2][n]
    out["points"] = pts_np
    out["labels"] = labels
    out["init_seed"] = np.int64(1234)
    out["fwd_seed"] = np.int64(4321)
    # G8: the four start vectors of one SSG forward, in call order sa1..sa4
    torch.manual_seed(4321)
    out["ssg/starts"] = np.stack([T.draw_start(2, n).numpy() for n in (1024, 1024, 256, 64)])
    save("g6_nets.npz", **out)


# ----------------------------------------------------------------------------- G6n the reference against itself
def g6_noise():
    """How far the REFERENCE moves against itself at the G6 size (B=2 x 1024) when only the thread count changes
    (8 vs 1 threads: another summation order in MKL / the BatchNorm reductions).  tests/test_parity_fullsize_gpu.py
    quotes these numbers next to the fp64 yardstick instead of asserting a remembered constant."""
    print("G6n reference self-noise, 8 threads vs 1")
    out = {}
    pts_np, labels = syn.kitti_batch(200, 2, 1024)
    pts = torch.from_numpy(pts_np)
    for tag, make_ref in [("ssg", lambda: R2.PointNet2SemSeg(13, 6)), ("msg", lambda: RefMSGSemSegFromReference(13, 6))]:
        runs = []
        for threads in (8, 1):
            torch.set_num_threads(threads)
            torch.manual_seed(1234)
            ref = make_ref()
            ref.drop1.p = 0.0
            ref.train()
            torch.manual_seed(4321)
            lp = ref(pts)
            T.seg_loss(lp, labels).backward()
            runs.append((lp.detach().numpy().copy(), {k: p.grad.numpy().copy() for k, p in ref.named_parameters()}))
        torch.set_num_threads(8)
        (la, ga), (lb, gb) = runs
        out[tag + "/log_probs_absdiff"] = np.float64(np.abs(la - lb).max())
        rel = {n: np.abs(ga[n] - gb[n]).max() / max(np.abs(ga[n]).max(), 1e-12) for n in ga
               if not ("conv" in n and n.endswith("bias") and n != "conv2.bias")}
        l2 = {n: abs(np.linalg.norm(ga[n].astype(np.float64)) - np.linalg.norm(gb[n].astype(np.float64))) /
              max(np.linalg.norm(ga[n].astype(np.float64)), 1e-12) for n in rel}
        out[tag + "/grad_relmax_worst"] = np.float64(max(rel.values()))
        out[tag + "/grad_l2_worst"] = np.float64(max(l2.values()))
        print("  %s: |dlog_probs| %.2e, worst grad relmax %.2e, worst grad L2 %.2e" % (
            tag, out[tag + "/log_probs_absdiff"], out[tag + "/grad_relmax_worst"], out[tag + "/grad_l2_worst"]))
    save("g6_noise.npz", **out)


# ----------------------------------------------------------------------------- G10 the other four zoo nets
def g10_zoo():
    """PointNet2ClsMsg / ClsSsg / PartSegSsg / PartSegMsg_one_hot (reference model/pointnet2.py:7-139), train mode,
    dropout off, B=2 x 1024: pins the oracle restatements of these nets and stores the reference's outputs."""
    print("G10 zoo nets")
    out = {}
    pts_np, _ = syn.kitti_batch(400, 2, 1024)
    xyz = torch.from_numpy(np.ascontiguousarray(pts_np[:, :3]))
    nrm = torch.from_numpy(np.ascontiguousarray(pts_np[:, 3:6]))
    cls = torch.zeros(2, 16)
    cls[0, 3] = 1.0
    cls[1, 11] = 1.0
    nets = [("cls_msg", lambda: R2.PointNet2ClsMsg(), lambda: T.RefClsMsg(dropout=0.0), (xyz,)),
            ("cls_ssg", lambda: R2.PointNet2ClsSsg(), lambda: T.RefClsSsg(dropout=0.0), (xyz,)),
            ("partseg_ssg", lambda: R2.PointNet2PartSegSsg(50), lambda: T.RefPartSegSsg(50, dropout=0.0), (xyz,)),
            ("partseg_msg", lambda: R2.PointNet2PartSegMsg_one_hot(50), lambda: T.RefPartSegMsgOneHot(50, dropout=0.0),
             (xyz, nrm, cls))]
    # ONE thread for both sides: with 8 threads the summation order of the BatchNorm / GEMM reductions follows the thread
    # timing, a handful of argmax / ReLU decisions of the tiny batch then fall differently from run to run and the gradient
    # check below passed or failed by luck (round 2 verdict).  The reference's own thread noise is measured by g10n.
    torch.set_num_threads(1)
    for tag, make_ref, make_orc, ins in nets:
        torch.manual_seed(77)
        ref = make_ref()
        for m in ref.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        torch.manual_seed(77)
        orc = make_orc()
        check(state_digest(ref) == state_digest(orc), tag + ": seeded init of oracle net equals reference net")
        res = {}
        for which, net in (("ref", ref), ("orc", orc)):
            net.train()
            torch.manual_seed(88)
            y = net(*ins)
            ys = y if isinstance(y, tuple) else (y,)
            gw = torch.randn(ys[0].shape, generator=torch.Generator().manual_seed(5))
            (ys[0] * gw).sum().backward()
            res[which] = ([t.detach().numpy() for t in ys], {k: p.grad.numpy().copy() for k, p in net.named_parameters()})
        for i, (a, b) in enumerate(zip(res["ref"][0], res["orc"][0])):
            d = np.abs(a - b).max()
            check(a.shape == b.shape and d <= 2e-5, "%s: oracle output %d within 2e-5 of reference (%.2e)" % (tag, i, d))
        names = sorted(res["ref"][1])
        worst = 0.0
        for n in names:
            a, b = res["ref"][1][n].astype(np.float64), res["orc"][1][n].astype(np.float64)
            if ("conv" in n and n.endswith("bias") and n != "conv2.bias") or (n.startswith("fc") and n.endswith("bias") and n != "fc3.bias"):
                continue          # zero gradient under the BatchNorm that follows
            worst = max(worst, abs(np.linalg.norm(a) - np.linalg.norm(b)) / max(np.linalg.norm(a), 1e-12))
        check(worst <= 1e-4, "%s: per-tensor gradient L2 norms within 1e-4 (%.2e; one thread on both sides: deterministic)" % (tag, worst))
        out[tag + "/n_out"] = np.int64(len(res["ref"][0]))
        for i, a in enumerate(res["ref"][0]):
            out["%s/shape/%d" % (tag, i)] = np.array(a.shape, np.int64)
            out["%s/out/%d" % (tag, i)] = a if a.size <= 4096 else a.reshape(-1)[::17].copy()
        out[tag + "/grad_names"] = np.array(names)
        out[tag + "/grad_l2"] = np.array([np.linalg.norm(res["ref"][1][n].astype(np.float64)) for n in names])
    out["points"] = pts_np
    out["cls_label"] = cls.numpy()
    out["init_seed"] = np.int64(77)
    out["fwd_seed"] = np.int64(88)
    out["gw_seed"] = np.int64(5)
    out["threads"] = np.int64(1)
    torch.set_num_threads(8)
    save("g10_zoo.npz", **out)


def _zoo_inputs():
    pts_np, _ = syn.kitti_batch(400, 2, 1024)
    xyz = torch.from_numpy(np.ascontiguousarray(pts_np[:, :3]))
    nrm = torch.from_numpy(np.ascontiguousarray(pts_np[:, 3:6]))
    cls = torch.zeros(2, 16)
    cls[0, 3] = 1.0
    cls[1, 11] = 1.0
    return xyz, nrm, cls


def g10_noise():
    """How far the REFERENCE's zoo nets move against themselves (the g10 setting: train mode, dropout off, B=2 x 1024) when
    only the thread count changes: the 1-thread run g10_zoo.npz stores against three 8-thread runs.  The GPU test of these
    nets (tests/test_parity_fullsize_gpu.py) allows twice this instead of a quoted constant."""
    print("G10n reference self-noise of the zoo nets, 8 threads (x3) vs 1")
    xyz, nrm, cls = _zoo_inputs()
    nets = [("cls_msg", lambda: R2.PointNet2ClsMsg(), (xyz,)), ("cls_ssg", lambda: R2.PointNet2ClsSsg(), (xyz,)),
            ("partseg_ssg", lambda: R2.PointNet2PartSegSsg(50), (xyz,)),
            ("partseg_msg", lambda: R2.PointNet2PartSegMsg_one_hot(50), (xyz, nrm, cls))]
    out = {}

    def run(make_ref, ins, threads):
        torch.set_num_threads(threads)
        torch.manual_seed(77)
        ref = make_ref()
        for m in ref.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        ref.train()
        torch.manual_seed(88)
        y = ref(*ins)
        ys = y if isinstance(y, tuple) else (y,)
        gw = torch.randn(ys[0].shape, generator=torch.Generator().manual_seed(5))
        (ys[0] * gw).sum().backward()
        return [t.detach().numpy().copy() for t in ys], {k: p.grad.numpy().astype(np.float64) for k, p in ref.named_parameters()}

    for tag, make_ref, ins in nets:
        base_y, base_g = run(make_ref, ins, 1)
        d_out, d_l2 = 0.0, 0.0
        for _ in range(3):
            y, g = run(make_ref, ins, 8)
            d_out = max(d_out, max(np.abs(a - b).max() / max(1.0, np.abs(a).max()) for a, b in zip(base_y, y)))
            for n in base_g:
                if ("conv" in n and n.endswith("bias") and n != "conv2.bias") or (n.startswith("fc") and n.endswith("bias") and n != "fc3.bias"):
                    continue
                d_l2 = max(d_l2, abs(np.linalg.norm(base_g[n]) - np.linalg.norm(g[n])) / max(np.linalg.norm(base_g[n]), 1e-12))
        out[tag + "/out_rel"] = np.float64(d_out)
        out[tag + "/grad_l2_rel"] = np.float64(d_l2)
        print("  %s: outputs move %.2e (of max(1, |out|)), per-tensor gradient L2 norms %.2e" % (tag, d_out, d_l2))
    torch.set_num_threads(8)
    save("g10_noise.npz", **out)


def g7_checkpoint():
    """Shipped checkpoint (eval mode): stores input + output only; the weights stay in the reference."""
    print("G7 shipped checkpoint, eval mode")
    path = os.path.join(REF, "checkpoints", "pointnet2-inview-0.55884-0001.pth")
    sd = torch.load(path, map_location="cpu")
    sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
    ref = R2.PointNet2SemSeg(19, 1)
    ref.load_state_dict(sd)
    ref.eval()
    orc = T.RefSSGSemSeg(19, 1)
    orc.load_state_dict(sd)
    orc.eval()
    pts = torch.from_numpy(syn.kitti_batch(300, 1, 2048, channels=4)[0])
    with torch.no_grad():
        torch.manual_seed(99)
        a = ref(pts).numpy()
        torch.manual_seed(99)
        b = orc(pts).numpy()
    check(np.abs(a - b).max() <= 2e-5, "checkpoint eval: oracle within 2e-5 (%.2e)" % np.abs(a - b).max())
    save("g7_checkpoint_eval.npz", points=pts.numpy(), log_probs=a, seed=np.int64(99),
         n_keys=np.int64(len(sd)), sha256=np.array(hashlib.sha256(open(path, "rb").read()).hexdigest()))


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g6n", "g7", "g10", "g10n"]
    table = dict(g1=g1_fps, g2=g2_ball, g3=g3_sqdist, g4=g4_interp, g5=g5_modules, g6=g6_nets, g7=g7_checkpoint, g10=g10_zoo, g6n=g6_noise,
                 g10n=g10_noise)
    for w in which:
        table[w]()
    print("all oracle checks passed")
