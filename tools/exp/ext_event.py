"""Does a hipStreamWaitEvent(..., hipEventWaitExternal) issued under stream capture become an event-wait node that orders the
replay behind work recorded on ANOTHER stream after the capture?  (torch.cuda.Event(external=True) is refused on ROCm.)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pointnet12_amd.parallel import _loaded_hip_runtime      # the runtime torch is linked against, never a second copy by bare name

hip = ctypes.CDLL(_loaded_hip_runtime())
dev = torch.device("cuda:0")
ev = ctypes.c_void_p()
assert hip.hipEventCreateWithFlags(ctypes.byref(ev), 0x2) == 0
comm = torch.cuda.Stream()
flag = torch.zeros(1, device=dev)
out = torch.zeros(1, device=dev)
side = torch.cuda.Stream()
# record once before capture so the event is valid
assert hip.hipEventRecord(ev, ctypes.c_void_p(comm.cuda_stream)) == 0
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    s = torch.cuda.current_stream()
    rc = hip.hipStreamWaitEvent(ctypes.c_void_p(s.cuda_stream), ev, 1)
    print("hipStreamWaitEvent(external) under capture ->", rc)
    out.copy_(flag)
torch.cuda.synchronize()
for trial in range(3):
    flag.zero_(); out.zero_()
    torch.cuda.synchronize()
    with torch.cuda.stream(comm):
        torch.cuda._sleep(200_000_000)          # ~100 ms
        flag.fill_(1.0)
        assert hip.hipEventRecord(ev, ctypes.c_void_p(comm.cuda_stream)) == 0
    g.replay()                                   # on the current stream, NOT ordered behind comm except through the node
    torch.cuda.synchronize()
    print("trial", trial, "out =", float(out), "(1.0: the replay waited for the comm stream)")
# control: same without the wait node
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    out.copy_(flag)
flag.zero_(); out.zero_(); torch.cuda.synchronize()
with torch.cuda.stream(comm):
    torch.cuda._sleep(200_000_000)
    flag.fill_(1.0)
g2.replay()
torch.cuda.synchronize()
print("control out =", float(out), "(0.0 expected)")
