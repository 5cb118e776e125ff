#!/usr/bin/env python3
"""Generate tests/golden/g8_train.npz (development container only) and pin oracle/train_ref.py while doing so.

* Loader: ``data_utils/SemKITTI_Loader.py`` cannot be imported here (cv2, redis and PIL-dependent siblings are
  absent), so its two pure functions ``pcd_normalize`` and ``pcd_jitter`` are compiled from the reference file's own
  syntax tree and executed unmodified; the ``__getitem__`` sequence (:93-113: normalise, jitter when training,
  ``np.random.choice(length, npoints, replace=True)``) is driven with ``np.random.seed`` so the committed vectors
  hold the draws as well.
* Adam: ``torch.optim.Adam`` exactly as semseg.py:106-111 builds it (lr 1e-3, betas (0.9, 0.999), eps 1e-08,
  weight_decay 1e-4 = the --decay_rate default), 12 steps over three tensors, seeded gradients.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_train.py
"""
import ast
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PN2_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

from oracle import train_ref as TR            # noqa: E402
from pointnet12_amd import synthetic as syn   # noqa: E402


def reference_loader_functions():
    path = os.path.join(REF, "data_utils", "SemKITTI_Loader.py")
    tree = ast.parse(open(path).read(), path)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("pcd_jitter", "pcd_normalize")]
    assert len(keep) == 2
    ns = {"np": np}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns["pcd_normalize"], ns["pcd_jitter"]


def raw_scan(seed, M):
    """A KITTI-like raw scan in metres with intensity in [0,1] (un-normalised), plus a few out-of-range rows."""
    rng = np.random.default_rng(seed)
    pts = syn.kitti_cloud(seed, M, M, 1)[:, :4].astype(np.float32)      # normalised coordinates
    raw = np.empty((M, 4), np.float32)
    raw[:, 0] = pts[:, 0] * 70
    raw[:, 1] = pts[:, 1] * 70
    raw[:, 2] = pts[:, 2] * 3
    raw[:, 3] = rng.uniform(0, 1, M).astype(np.float32)
    raw[:5] = [[80, -75, 4, 1.2], [-71, 70.00001, -3.5, -0.2], [0, 0, 0, 0.5], [69.99999, 1e-30, 3, 1], [1e-40, -1e-40, 0, 0]]
    label = rng.integers(0, 19, M).astype(np.int32)
    return raw, label


def main():
    out = {}
    ref_normalize, ref_jitter = reference_loader_functions()
    cases = []
    for tag, seed, M, npoints, train in (("train_a", 11, 5000, 4096, True), ("train_small", 12, 700, 2048, True),
                                         ("eval", 13, 3000, 1024, False)):
        raw, label = raw_scan(seed, M)
        np.random.seed(1000 + seed)
        pcd = ref_normalize(raw)                                   # SemKITTI_Loader.py:95
        if train:
            pcd = ref_jitter(pcd)                                  # :96-97
        choice = np.random.choice(pcd.shape[0], npoints, replace=True)   # :110-111
        pts, lab = pcd[choice], label[choice]
        np.random.seed(1000 + seed)
        mine_pts, mine_lab, noise, mine_choice = TR.prepare_cloud(raw, label, npoints, train)
        assert pts.dtype == np.float32 and (mine_choice == choice).all()
        assert (mine_pts.view(np.uint32) == pts.view(np.uint32)).all(), tag
        assert (mine_lab == lab).all()
        print("  ok: loader", tag, "bit-equal")
        out.update({tag + "/raw": raw, tag + "/label": label, tag + "/np_seed": np.int64(1000 + seed),
                    tag + "/train": np.bool_(train), tag + "/points": pts, tag + "/labels": lab,
                    tag + "/choice": choice})
        cases.append(tag)
    out["loader_cases"] = np.array(cases)

    # ---- Adam
    torch.manual_seed(7)
    shapes = [(64, 9, 1, 1), (64,), (13, 128, 1)]
    params = [torch.nn.Parameter(torch.randn(s) * 0.3) for s in shapes]
    opt = torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.999), eps=1e-08, weight_decay=1e-4)
    flat0 = torch.cat([p.detach().reshape(-1) for p in params]).numpy().copy()
    p_np, m_np, v_np = flat0.copy(), np.zeros_like(flat0), np.zeros_like(flat0)
    grads, after = [], []
    for t in range(1, 13):
        g = torch.cat([torch.randn(s).reshape(-1) for s in shapes]) * (10.0 ** torch.randint(-4, 1, (1,)).item())
        if t == 5:
            g = torch.zeros_like(g)
        grads.append(g.numpy().copy())
        off = 0
        for p in params:
            p.grad = g[off:off + p.numel()].view_as(p).clone()
            off += p.numel()
        if t == 7:
            opt.param_groups[0]["lr"] = 5e-4                       # a scheduler step (semseg.py:113)
        opt.step()
        after.append(torch.cat([p.detach().reshape(-1) for p in params]).numpy().copy())
        TR.adam_step(p_np, grads[-1], m_np, v_np, t, lr=opt.param_groups[0]["lr"], weight_decay=1e-4)
        err = np.abs(p_np - after[-1]).max() / np.abs(after[-1]).max()
        assert err < 2e-7, (t, err)
    st = opt.state[params[0]]
    n0 = params[0].numel()
    assert np.abs(m_np[:n0] - st["exp_avg"].reshape(-1).numpy()).max() < 1e-7
    print("  ok: Adam 12 steps, max rel err %.2e" % err)
    out.update({"adam/param0": flat0, "adam/grads": np.stack(grads), "adam/after": np.stack(after),
                "adam/lr": np.array([1e-3] * 6 + [5e-4] * 6), "adam/weight_decay": np.float64(1e-4),
                "adam/exp_avg": torch.cat([opt.state[p]["exp_avg"].reshape(-1) for p in params]).numpy(),
                "adam/exp_avg_sq": torch.cat([opt.state[p]["exp_avg_sq"].reshape(-1) for p in params]).numpy()})
    path = os.path.join(ROOT, "tests", "golden", "g8_train.npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
