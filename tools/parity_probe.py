#!/usr/bin/env python3
"""GPU: where does the HIP path's distance to the fp64 evaluation come from?  Runs the full-size MSG / SSG parity case of
tests/test_parity_fullsize_gpu.py under a few switches and prints the yardstick numbers side by side.

    python tools/parity_probe.py [msg|ssg] [variant ...]      variants: default nofact noseg
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch                                   # noqa: E402

import test_parity_fullsize_gpu as P           # noqa: E402
from pointnet12_amd import pointnet_util as U  # noqa: E402
from pointnet12_amd import synthetic as syn    # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "msg"
variants = sys.argv[2:] or ["default", "nofact", "noseg"]
dev = torch.device("cuda:0")
pts_np, lab_np = syn.kitti_batch(0, 16, 4096)
pts, labels = torch.from_numpy(pts_np), torch.from_numpy(lab_np)
net, orc = P._nets(kind, dev)
o32 = P._run_oracle(orc, pts, labels, torch.float32)
o64 = P._run_oracle(orc, pts, labels, torch.float64)
for v in variants:
    U.FACTORISE_MIN_FEATURES = 10 ** 9 if v == "nofact" else 32
    U.GATHER_BACKWARD = v != "noseg"
    hip = P._run_hip(net, pts, labels, dev)
    r = P._compare("probe_%s_%s" % (kind, v), hip, o32, o64)
    print(v, {k: (("%.3g" % x) if isinstance(x, float) else x) for k, x in r.items()})
