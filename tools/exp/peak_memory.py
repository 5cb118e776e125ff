"""Peak device memory of one eager cfg5 MSG step (B = 8 x 65 536, npoint x16) with the MSG scale outputs written in place into
one concatenated matrix (default) and through torch.cat (PN2_MSG_CONCAT_IN_PLACE=0).  ADVICE r3 (low): in place, every scale's
arg-max / masked-gradient buffers take the pitch of the WHOLE matrix.  Run once per setting (the switch is read at import)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pointnet12_amd import parallel, pointnet2 as M, synthetic as syn
from pointnet12_amd.loss import nll_loss

dev = torch.device("cuda:0")
B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 65536
pts_np, lab_np = syn.kitti_batch(0, B, N)
pts, lab = torch.from_numpy(pts_np).to(dev), torch.from_numpy(lab_np).to(dev)
torch.manual_seed(0)
net = M.PointNet2SemSegMsg(13, 6, npoint_scale=16).to(dev).train()
bucket = parallel.FlatGradBucket(net, direct=True)
for it in range(2):
    torch.cuda.reset_peak_memory_stats()
    bucket.zero()
    lp = net(pts)
    nll_loss(lp.reshape(-1, 13), lab.reshape(-1)).backward()
    torch.cuda.synchronize()
print("PN2_MSG_CONCAT_IN_PLACE=%s  B=%d x %d  peak allocated %.2f GB  reserved %.2f GB" % (
    os.environ.get("PN2_MSG_CONCAT_IN_PLACE", "1"), B, N, torch.cuda.max_memory_allocated() / 1e9, torch.cuda.max_memory_reserved() / 1e9))
