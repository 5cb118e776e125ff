#!/usr/bin/env python3
"""Phase stamps of the pruned FPS kernel (debug build: PN2_FPS_DBG, libpn2_dbg.so via PN2_LIB_PATH)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnet12_amd import _lib, pointnet_util as U, synthetic as syn

N = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
dev = torch.device("cuda:0")
pts, _ = syn.kitti_batch(1, 1, min(N, 65536))
xyz = torch.from_numpy(pts[:, :3].transpose(0, 2, 1).copy()).to(dev)[:, :N].contiguous()
if "--uniform" in sys.argv:
    xyz = torch.rand(1, N, 3, device=dev) * torch.tensor([2.0, 2.0, 0.2], device=dev)
start = torch.zeros(1, dtype=torch.int64, device=dev)
U.farthest_point_sample(xyz, 1024, start)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(1024 * 16 * 6, np.uint64)
rc = raw.pn2_fps_debug_read(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
assert rc == 0, rc
d = buf.reshape(1024, 16, 6).astype(np.int64)
t0, tt, tr, ti, tb, g = [d[:, :, k] for k in range(6)]
groups, rows = g & 0xFFFFFFFF, g >> 32
act = tr > 0
it_len = (tb.max(1)[1:] - tb.max(1)[:-1])
print("cycles per iteration (barrier to barrier): median %d  mean %d" % (np.median(it_len), it_len.mean()))
print("test phase (t_test - t0): median %d" % np.median(tt - t0))
rows_ph = np.where(act, tr - tt, 0)
tie_ph = np.where(act, ti - tr, 0)
print("row phase of the slowest wave: median %d mean %d;  tie phase of the slowest wave: median %d mean %d" %
      (np.median(rows_ph.max(1)), rows_ph.max(1).mean(), np.median(tie_ph.max(1)), tie_ph.max(1).mean()))
print("active waves per iteration: mean %.2f;  touched rows: total mean %.1f, max wave mean %.1f;  touched groups: max wave mean %.2f" %
      (act.sum(1).mean(), rows.sum(1).mean(), rows.max(1).mean(), groups.max(1).mean()))
tail = np.where(act, tb - ti, 0)
print("after-tie to barrier exit of slowest: median %d" % np.median((tb.max(1) - np.where(act, ti, 0).max(1))[act.any(1)]))
for lo, hi in ((0, 32), (32, 128), (128, 512), (512, 1024)):
    sl = slice(lo, hi)
    print("iterations %4d..%4d: %.0f cycles/iter, rows max-wave %.1f, groups max-wave %.2f, active waves %.2f" %
          (lo, hi, it_len[lo:hi - 1].mean(), rows[sl].max(1).mean(), groups[sl].max(1).mean(), act[sl].sum(1).mean()))
