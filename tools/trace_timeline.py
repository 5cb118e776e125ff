#!/usr/bin/env python3
"""One graph-replayed step of a rocprofv3 --kernel-trace CSV as a timeline.

    python tools/trace_timeline.py <kernel_trace.csv> [out.txt]

The launches between the last two loss kernels are one period.  Prints every launch in start order (offset, duration,
queue, how many other kernels were running when it started, the idle time of the whole device just before it) and a
summary: union-busy time, idle time, the idle time charged to the kernel that follows it, per-kernel-name totals of
"exposed" time (time during which the kernel was the only one running)."""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n)
    return re.sub(r"\s+", " ", n)[:64]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [int(r["Start_Timestamp"]) for r in rows if "nll_fwd_kernel" in r["Kernel_Name"]]
    last = [r for r in rows if marks[-2] <= int(r["Start_Timestamp"]) < marks[-1]]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    t0 = int(last[0]["Start_Timestamp"])
    iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in last]
    # exposed time per kernel: sweep over the elementary intervals
    pts = sorted(set([a for a, _, _, _ in iv] + [b for _, b, _, _ in iv]))
    exposed = collections.defaultdict(float)
    idle_before = collections.defaultdict(float)
    busy = idle = 0
    active = []
    k = 0
    order = sorted(range(len(iv)), key=lambda i: iv[i][0])
    for a, b in zip(pts[:-1], pts[1:]):
        while k < len(order) and iv[order[k]][0] <= a:
            active.append(order[k]); k += 1
        active = [i for i in active if iv[i][1] > a]
        if not active:
            idle += b - a
            nxt = iv[order[k]][2] if k < len(order) else "-"
            idle_before[nxt] += b - a
        else:
            busy += b - a
            if len(active) == 1:
                exposed[iv[active[0]][2]] += b - a
    print("period: %d kernels, span %.1f us, busy %.1f us, idle %.1f us" % (len(iv), (pts[-1] - pts[0]) / 1e3, busy / 1e3, idle / 1e3), file=out)
    print("\nexposed (sole running) time and idle time charged to the following kernel, by kernel name:", file=out)
    cnt = collections.Counter(n for _, _, n, _ in iv)
    tot = collections.defaultdict(float)
    for a, b, n, _ in iv:
        tot[n] += b - a
    for n in sorted(tot, key=lambda n: -(exposed[n] + idle_before[n])):
        print("  %-64s x%-3d total %8.1f  exposed %8.1f  idle-before %7.1f" % (n, cnt[n], tot[n] / 1e3, exposed[n] / 1e3, idle_before[n] / 1e3), file=out)
    print("\ntimeline:", file=out)
    ends = []
    for a, b, n, q in iv:
        running = sum(1 for e in ends if e > a)
        gap = a - max(ends) if ends and max(ends) < a else 0
        print("  %9.1f  %7.1f us  q%-3s run%-2d gap %5.1f  %s" % ((a - t0) / 1e3, (b - a) / 1e3, q, running, gap / 1e3, n), file=out)
        ends.append(b)


if __name__ == "__main__":
    main()
