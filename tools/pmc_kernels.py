#!/usr/bin/env python3
"""Join the passes of tools/pmc_kernels.sh: per kernel (name, grid) the mean of every counter over its dispatches, the mean
duration from the kernel trace, and the derived clock / matrix-pipe utilisation."""
import collections
import csv
import glob
import os
import re
import sys

out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sorted(glob.glob(os.path.join(out, "p*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            key = (re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:60], r.get("Grid_Size", ""))
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE"):     # once per pass and dispatch
                dur[key].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3)
for key in sorted(agg, key=lambda k: -sum(dur.get(k, [0])) ):
    c = {k: sum(v) / len(v) for k, v in agg[key].items()}
    us = sorted(dur[key])[len(dur[key]) // 2] if dur.get(key) else float("nan")
    if us < 15:
        continue
    line = "%-60s grid %-8s %8.1f us" % (key[0], key[1], us)
    if "GRBM_GUI_ACTIVE" in c and us == us:
        clk = c["GRBM_GUI_ACTIVE"] / 8 / us / 1e3        # GHz
        line += "  clk %.2f GHz" % clk
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            line += "  mfma_busy %.2f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8) / 1024)
    print(line)
    print("     " + "  ".join("%s=%.3g" % (k, v) for k, v in sorted(c.items())))
