"""The 61 us cadence, isolated?  One captured graph: FPS (B=16 x 4096 -> 1024) on a side branch, a chain of small kernels on the main
branch (a mix: torch elementwise, a pn2 GEMM, a pn2 reduction).  Run under rocprofv3 --kernel-trace and read the chain's start times."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import pointnet12_amd.pointnet_util as U

dev = torch.device("cuda:0")
torch.manual_seed(0)
xyz = torch.rand(16, 4096, 3, device=dev)
start = torch.zeros(16, dtype=torch.int64, device=dev)
a = torch.zeros(1 << 20, device=dev)
x = torch.randn(65536, 128, device=dev)
w = torch.randn(128, 128, device=dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "ingraph"


def chain():
    for i in range(6):
        a.add_(1.0)
    y = x @ w
    for i in range(6):
        a.mul_(1.0001)
    return y


side = torch.cuda.Stream(device=dev)
for _ in range(2):
    U.farthest_point_sample(xyz, 1024, start=start); chain()
torch.cuda.synchronize()
if mode == "ingraph":
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            U.farthest_point_sample(xyz, 1024, start=start)
        chain()
        main.wait_stream(side)
    for _ in range(6):
        g.replay()
else:                        # two graphs, two streams
    g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        U.farthest_point_sample(xyz, 1024, start=start)
    with torch.cuda.graph(g2):
        chain()
    s2 = torch.cuda.Stream(device=dev)
    for _ in range(6):
        with torch.cuda.stream(side):
            g1.replay()
        with torch.cuda.stream(s2):
            g2.replay()
        torch.cuda.synchronize()
torch.cuda.synchronize()
