#!/usr/bin/env python3
"""Join the passes of tools/exp/pmc_exp.sh per case: mean of every counter over the dispatches, derived clock and pipe use."""
import collections, csv, glob, os, re, sys
out = sys.argv[1]
cases = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sorted(glob.glob(os.path.join(out, "p*_c*"))):
    if not os.path.isdir(d):
        continue
    case = d.rsplit("_c", 1)[1]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            cases[case][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"):
                dur[case].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3)
for case in sorted(cases, key=int):
    c = {k: sum(v) / len(v) for k, v in cases[case].items()}
    us = sorted(dur[case])[len(dur[case]) // 2]
    line = "case %-3s %8.1f us" % (case, us)
    if "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        line += "  clk %.2f GHz" % (cyc / us / 1e3)
        if "SQ_INSTS_MFMA" in c:
            line += "  mfma_issue_frac(64cyc) %.3f" % (c["SQ_INSTS_MFMA"] * 64 / (cyc * 1024))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            line += "  mfma_busy %.3f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / cyc / 1024)
    if "SQ_WAVE_CYCLES" in c:
        w = c["SQ_WAVE_CYCLES"]
        line += "  wait_any %.2f wait_inst %.2f active %.2f" % (c.get("SQ_WAIT_ANY", 0) / w, c.get("SQ_WAIT_INST_ANY", 0) / w, c.get("SQ_ACTIVE_INST_ANY", 0) / w)
    print(line)
    print("     " + "  ".join("%s=%.4g" % (k, v) for k, v in sorted(c.items())))
