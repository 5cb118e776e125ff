#!/usr/bin/env python3
"""Development container only: step time of the oracle in "aten" geometry mode (what bench.py's cpu_baseline times)
against the REFERENCE itself on the same batch, same thread count.  BASELINE.md section 3 asks for +-10 %.

    PYTHONDONTWRITEBYTECODE=1 python tools/check_cpu_port_timing.py [ssg|msg] [B]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, os.environ.get("PN2_REFERENCE", "/root/reference"))
sys.path.insert(0, ROOT)
from model import pointnet2 as R2                 # noqa: E402  (the reference)
from oracle import torch_ref as T                 # noqa: E402
from pointnet12_amd import synthetic as syn       # noqa: E402
from tools.make_golden import RefMSGSemSegFromReference   # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "ssg"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
torch.set_num_threads(8)
pts_np, lab = syn.kitti_batch(0, B, 4096)
pts, lab = torch.from_numpy(pts_np), torch.from_numpy(lab)


def timed(net):
    net.train()
    out = []
    for i in range(3):
        net.zero_grad()
        torch.manual_seed(1234)
        t0 = time.perf_counter()
        T.seg_loss(net(pts), lab).backward()
        out.append(time.perf_counter() - t0)
    return float(np.median(out[1:])), out


torch.manual_seed(0)
ref = R2.PointNet2SemSeg(13, 6) if kind == "ssg" else RefMSGSemSegFromReference(13, 6)
t_ref, all_ref = timed(ref)
T.set_geometry("aten")
torch.manual_seed(0)
orc = T.RefSSGSemSeg(13, 6) if kind == "ssg" else T.RefMSGSemSeg(13, 6)
t_orc, all_orc = timed(orc)
print("%s B=%d 8 threads: reference %.2f s/step %s, oracle(aten) %.2f s/step %s, ratio %.3f" % (
    kind, B, t_ref, np.round(all_ref, 2), t_orc, np.round(all_orc, 2), t_orc / t_ref))
