#!/usr/bin/env python3
"""Pretty-print a bench.py JSON line: step time, roofline object, per-entry-point device time."""
import json
import sys

d = json.load(open(sys.argv[1]))
print(d["config"]["workload"][:40], "| %.3f ms/step | %.2f M pts/s" % (d["ms_per_step"], d["value"] / 1e6))
print("   roofline:", d["roofline"])
for k, v in list((d.get("kernels") or {}).items())[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print("   %-24s %7.3f ms  n=%3d  %8.1f GF %9.1f MB" % (k, v["ms_per_step"], v["launches_per_step"], v["gflop_per_step"], v["mb_per_step"]))
