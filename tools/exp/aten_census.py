#!/usr/bin/env python3
"""Which ATen operators still launch kernels inside one eager training step, and from which line of the package
(TorchDispatchMode + the Python stack): python tools/exp/aten_census.py [msg|ssg]"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from pointnet12_amd import synthetic as syn, parallel

wl = sys.argv[1] if len(sys.argv) > 1 else "ssg"
dev = torch.device("cuda:0")
net = bench.build_net(wl, dev)
pts_np, lab_np = syn.kitti_batch(0, 16, 4096)
pts, lab = torch.from_numpy(pts_np).to(dev), torch.from_numpy(lab_np).to(dev)
bucket = parallel.FlatGradBucket(net)
step = bench.make_step(wl, net, pts, lab, bucket)
for _ in range(3):
    step()
torch.cuda.synchronize()
SKIP = ("aten.view", "aten.empty", "aten.as_strided", "aten.detach", "aten.alias", "aten.reshape", "aten._unsafe_view", "aten.slice", "aten.select",
        "aten.transpose", "aten.permute", "aten.expand", "aten.unsqueeze", "aten.squeeze", "aten.t.", "aten.stride", "aten.size", "aten.is_", "aten.sym_",
        "aten.lift_fresh", "aten._local_scalar", "aten.narrow", "aten.unbind", "aten.split", "aten.new_empty", "aten.empty_like", "aten.numel")
count = collections.Counter()


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(name.startswith(s) for s in SKIP):
            where = "(autograd engine)"
            for fr in reversed(traceback.extract_stack()):
                if ("pointnet12_amd" in fr.filename or fr.filename.endswith("bench.py")) and "aten_census" not in fr.filename:
                    where = "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
                    break
            shp = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), ())
            count[(name, where, shp)] += 1
        return func(*args, **(kwargs or {}))


with Census():
    step()
torch.cuda.synchronize()
for (name, where, shp), n in sorted(count.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print("%-34s x%-3d %-22s %s" % (name, n, shp, where))
