import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pointnet12_amd import _lib
from pointnet12_amd._lib import ptr as p
lib = _lib.load(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
for G, K, C in [(8192, 128, 128), (8192, 64, 128), (8192, 32, 64), (2048, 128, 256), (16384, 32, 64)]:
    Y = torch.randn(G * K, C, device=dev); aff = torch.zeros(4 * C, device=dev); aff[C:2*C] = 1
    out = torch.empty(G, C, device=dev); arg = torch.empty(G, C, device=dev, dtype=torch.int32)
    f = lambda: lib.pn2_bn_relu_max(p(Y), C, p(aff), G, K, C, p(out), C, p(arg), st)
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print("G=%6d K=%4d C=%4d  %7.1f us  %6.0f GB/s" % (G, K, C, ms * 1e3, G * K * C * 4 / ms / 1e6))
