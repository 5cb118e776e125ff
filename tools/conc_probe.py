import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pointnet12_amd import pointnet_util as U, synthetic as syn
dev = torch.device("cuda:0")
pts, _ = syn.kitti_batch(1, 16, 4096)
xyz = torch.from_numpy(pts[:, :3].transpose(0, 2, 1).copy()).to(dev)
start = torch.zeros(16, dtype=torch.int64, device=dev)
a = torch.randn(8192, 8192, device=dev); b = torch.randn(8192, 8192, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def fps():
    with torch.cuda.stream(s1):
        U.farthest_point_sample(xyz, 1024, start)
def mm():
    with torch.cuda.stream(s2):
        for _ in range(4): torch.mm(a, b)
def timeit(fns):
    for f in fns: f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in fns: f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
print("fps alone %.2f ms" % timeit([fps]))
print("mm alone  %.2f ms" % timeit([mm]))
print("fps then mm (two streams) %.2f ms" % timeit([fps, mm]))
print("mm then fps (two streams) %.2f ms" % timeit([mm, fps]))
xyz2 = xyz.clone(); start2 = start.clone()
def fps2():
    with torch.cuda.stream(s2):
        U.farthest_point_sample(xyz2, 1024, start2)
print("fps || fps (two streams) %.2f ms" % timeit([fps, fps2]))
x = torch.randn(1 << 26, device=dev)
def ew():
    with torch.cuda.stream(s2):
        for _ in range(20): x.mul_(1.0001)
print("ew alone %.2f ms" % timeit([ew]))
print("fps || ew %.2f ms" % timeit([fps, ew]))
print(os.environ.get("GPU_MAX_HW_QUEUES"), os.environ.get("AMD_SERIALIZE_KERNEL"), os.environ.get("HIP_LAUNCH_BLOCKING"), os.environ.get("CUDA_LAUNCH_BLOCKING"))
