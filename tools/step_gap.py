#!/usr/bin/env python3
"""From a rocprofv3 kernel-trace CSV of bench.py: time between the end of a step's last GEMM and the next step's first
GEMM (the region where only small kernels / the prefetch branch run)."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gemm = [r for r in rows if "gemm_" in r["Kernel_Name"] or "wgrad_skinny" in r["Kernel_Name"]]
# step boundaries: gaps between consecutive GEMM launches larger than 150 us
gaps = []
prev_end = int(gemm[0]["End_Timestamp"])
for r in gemm[1:]:
    s = int(r["Start_Timestamp"])
    if s - prev_end > 150000:
        gaps.append((s - prev_end) / 1e3)
    prev_end = max(prev_end, int(r["End_Timestamp"]))
print("GEMM-free gaps > 150 us:", " ".join("%.0f" % g for g in gaps[-12:]))
