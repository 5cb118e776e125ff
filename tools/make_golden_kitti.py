#!/usr/bin/env python3
"""Generate tests/golden/g9_kitti.npz (development container only): the reference's own scan reader on synthetic
.bin / .label files.

``data_utils/kitti_utils.py`` cannot be imported here (cv2), so the five methods of ``Semantic_KITTI_Utils`` that the
read path consists of -- ``get``, ``set_filter``, ``hv_in_range``, ``box_in_range``, ``points_basic_filter`` -- are
compiled from the reference file's own syntax tree into a bare class and run unmodified; ``learning_map`` is read
from the reference's ``config/semantic-kitti.yaml``.  Stored: the two input files' contents, the map, and what
``get`` returned for both subsets.  pointnet12_amd/kitti.py is asserted bit-equal while doing so.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_kitti.py
"""
import ast
import os
import sys
import tempfile

import numpy as np
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PN2_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

from pointnet12_amd import kitti, synthetic as syn   # noqa: E402

WANT = ("get", "set_filter", "hv_in_range", "box_in_range", "points_basic_filter")


def reference_reader():
    path = os.path.join(REF, "data_utils", "kitti_utils.py")
    tree = ast.parse(open(path).read(), path)
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Semantic_KITTI_Utils"][0]
    cls.body = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in WANT]
    assert len(cls.body) == len(WANT)
    ns = {"np": np, "os": os}
    exec(compile(ast.Module(body=[cls], type_ignores=[]), path, "exec"), ns)
    return ns["Semantic_KITTI_Utils"]


def main():
    learning_map = yaml.safe_load(open(os.path.join(REF, "config", "semantic-kitti.yaml")))["learning_map"]
    Reader = reference_reader()
    rng = np.random.default_rng(2026)
    M = 6000
    n = syn.kitti_cloud(77, M, M, 1)[:, :4].astype(np.float32)
    raw = np.stack([n[:, 0] * 70, n[:, 1] * 70, n[:, 2] * 3, n[:, 3] / 2 + 0.5], 1).astype(np.float32)
    # the synthetic cloud only spans the camera's azimuth: add a ring all around, steep rays and the exact FOV borders
    ang = rng.uniform(-np.pi, np.pi, 1500)
    rad = rng.uniform(2, 60, 1500)
    ring = np.stack([rad * np.cos(ang), rad * np.sin(ang), rng.uniform(-25, 25, 1500), rng.uniform(0, 1, 1500)], 1)
    edge = []
    for deg in (40.0, -40.0):
        for eps in (-1e-6, 0.0, 1e-6):
            a = np.deg2rad(deg) + eps
            edge.append([10 * np.cos(a), 10 * np.sin(a), 0.0, 0.5])
    for deg in (20.0, -20.0):
        for eps in (-1e-6, 0.0, 1e-6):
            a = np.deg2rad(deg) + eps
            edge.append([10 * np.cos(a), 0.0, 10 * np.sin(a), 0.5])       # atan2(z, d) with d the 3-D range
    edge.append([0.0, 0.0, 0.0, 0.0])
    raw = np.concatenate([raw, ring.astype(np.float32), np.array(edge, np.float32)], 0)
    keys = np.array(sorted(learning_map))
    sem = keys[rng.integers(0, len(keys), raw.shape[0])].astype(np.uint32)
    inst = rng.integers(0, 300, raw.shape[0]).astype(np.uint32)
    label_file = (sem | (inst << 16)).astype(np.uint32)
    out = {"bin": raw, "label": label_file, "map_keys": keys.astype(np.int64),
           "map_values": np.array([learning_map[k] for k in keys], np.int64)}
    with tempfile.TemporaryDirectory() as tmp:
        seq = os.path.join(tmp, "sequences", "04")
        os.makedirs(os.path.join(seq, "velodyne"))
        os.makedirs(os.path.join(seq, "labels"))
        fv, fl = os.path.join(seq, "velodyne", "000003.bin"), os.path.join(seq, "labels", "000003.label")
        raw.tofile(fv)
        label_file.tofile(fl)
        for subset in ("all", "inview"):
            r = Reader.__new__(Reader)                       # no __init__: it opens calibration files and cv2 objects
            r.root, r.subset, r.num_classes = tmp + "/", subset, 19
            r.length = {"04": 270}
            r.learning_map = learning_map
            pts, lab = r.get("04", 3)
            mine_p, mine_l = kitti.read_scan(fv, fl, learning_map, subset)
            assert pts.dtype == np.float32 and lab.dtype == mine_l.dtype, (pts.dtype, lab.dtype, mine_l.dtype)
            assert pts.shape == mine_p.shape and (pts.view(np.uint32) == mine_p.view(np.uint32)).all(), subset
            assert (lab == mine_l).all()
            print("  ok: %-6s %5d of %5d points kept, classes %d..%d" % (subset, len(pts), len(raw), lab.min(), lab.max()))
            out[subset + "/points"], out[subset + "/labels"] = pts, lab
    path = os.path.join(ROOT, "tests", "golden", "g9_kitti.npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
