"""Which form of an event record INSIDE a captured graph does this HIP runtime accept (two-bucket all-reduce, VERDICT r3 #7b)?
Tries hipEventRecordWithFlags(..., hipEventRecordExternal) under torch's stream capture with events of several creation flags,
with and without a record before the capture, and on the system runtime's entry point for comparison."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pointnet12_amd.parallel import _loaded_hip_runtime

hip = ctypes.CDLL(_loaded_hip_runtime())
print("runtime:", _loaded_hip_runtime(), "has hipEventRecordWithFlags:", hasattr(hip, "hipEventRecordWithFlags"))
ver = ctypes.c_int()
hip.hipRuntimeGetVersion(ctypes.byref(ver))
print("hipRuntimeGetVersion:", ver.value)
dev = torch.device("cuda:0")
x = torch.zeros(4, device=dev)
for flags in (0x0, 0x2, 0x1, 0x3):
    for pre in (False, True):
        ev = ctypes.c_void_p()
        rc = hip.hipEventCreateWithFlags(ctypes.byref(ev), flags)
        if rc:
            print("create flags %#x -> %d" % (flags, rc))
            continue
        if pre:
            hip.hipEventRecord(ev, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        res = {}
        try:
            with torch.cuda.graph(g):
                x.add_(1.0)
                s = torch.cuda.current_stream().cuda_stream
                res["ext"] = hip.hipEventRecordWithFlags(ev, ctypes.c_void_p(s), 1)
                x.add_(1.0)
        except Exception as e:
            res["exc"] = repr(e)[:120]
        print("event flags %#x, recorded before capture %s -> hipEventRecordWithFlags(External) under capture = %s" % (flags, pre, res))
        torch.cuda.synchronize()
