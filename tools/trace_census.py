#!/usr/bin/env python3
"""Per-step kernel census of a rocprofv3 --kernel-trace CSV of graph-replayed bench steps.

    python tools/trace_census.py gpurun_out/prof_f/msg_kernel_trace.csv [steps_in_trace_tail=2]

Takes the launches between the last two loss kernels as one step and prints, per kernel name, launches per step,
summed duration per step, and the span / union-busy time of one step."""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n)
    return re.sub(r"\s+", " ", n)[:100]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # one period = the launches between the last two loss kernels (one per step; the phase does not matter)
    marks = [int(r["Start_Timestamp"]) for r in rows if "nll_fwd_kernel" in r["Kernel_Name"]]
    if len(marks) < 3:
        print("fewer than three steps in the trace")
        return
    last = [r for r in rows if marks[-2] <= int(r["Start_Timestamp"]) < marks[-1]]
    period = len(last)
    t0 = min(int(r["Start_Timestamp"]) for r in last)
    t1 = max(int(r["End_Timestamp"]) for r in last)
    ev = sorted([(int(r["Start_Timestamp"]), 1) for r in last] + [(int(r["End_Timestamp"]), -1) for r in last])
    busy, cur, prev = 0, 0, t0
    for t, d in ev:
        if cur > 0:
            busy += t - prev
        cur += d
        prev = t
    print("kernels per period %d   span %.1f us   union busy %.1f us" % (period, (t1 - t0) / 1e3, busy / 1e3))
    agg = collections.defaultdict(lambda: [0, 0])
    for r in last:
        a = agg[short(r["Kernel_Name"])]
        a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a[1] += 1
    for n, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print("%9.1f us %4d  %s" % (d / 1e3, c, n))


if __name__ == "__main__":
    main()
