"""Does a running FPS launch delay the dispatch of small kernels on ANOTHER stream?  (rocprofv3 trace of the captured MSG step: the
four kernels behind the loss start 61 us apart while fps_kernel<512, 8> runs on the other queue.)  Eager, two torch streams, HIP events."""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import pointnet12_amd.pointnet_util as U

dev = torch.device("cuda:0")
torch.manual_seed(0)
xyz = torch.rand(16, 4096, 3, device=dev)
a = torch.zeros(1024, device=dev)
sA, sB = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def chain(n=8):
    for _ in range(n):
        a.add_(1.0)


def run(with_fps, graph=False):
    torch.cuda.synchronize()
    e0, e1, f0, f1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    if with_fps:
        with torch.cuda.stream(sA):
            f0.record()
            U.farthest_point_sample(xyz, 1024)
            f1.record()
        time.sleep(0.0002)            # (the FPS launch is resident before the chain is queued)
    with torch.cuda.stream(sB):
        e0.record()
        chain()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3, (f0.elapsed_time(f1) * 1e3 if with_fps else 0.0)


for _ in range(3):
    run(True); run(False)
for w in (False, True, False, True):
    r = [run(w) for _ in range(5)]
    print("fps running" if w else "alone      ", "chain of 8 tiny kernels: us", [round(x[0], 1) for x in r], "fps us", [round(x[1], 1) for x in r])

# the same inside one captured graph: a fork at the top, FPS on one branch, the chain on the other
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream(device=dev)
for nfps in (0, 1):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)
        if nfps:
            with torch.cuda.stream(side):
                U.farthest_point_sample(xyz, 1024)
        chain(8)
        main.wait_stream(side)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print("graph, %d fps branch + chain of 8: %.1f us per replay" % (nfps, e0.elapsed_time(e1) * 100))
