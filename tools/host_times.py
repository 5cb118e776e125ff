import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from pointnet12_amd import graph as G
orig = G.GraphedStep.__call__
times = []
def timed(self):
    t0 = time.perf_counter()
    r = orig(self)
    times.append((time.perf_counter() - t0) * 1e3)
    return r
G.GraphedStep.__call__ = timed
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-roofline", "--steps", "20", "--warmup", "5"] + sys.argv[1:]
bench.main()
print("host ms per replay call:", " ".join("%.2f" % t for t in times), file=sys.stderr)
