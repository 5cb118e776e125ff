#!/usr/bin/env python3
"""Step-level reading of a rocprofv3 --kernel-trace CSV of `bench.py` (graph replay).

    python3 tools/step_timeline.py <..._kernel_trace.csv> [--anchor nll_fwd_kernel] [--dump N]

Splits the trace into steps at every launch of the anchor kernel (one per step) and prints, per step: wall time between
anchors, kernel count, kernels per hardware queue, idle time (no kernel running) -- then, over the last full step (or the
N-th from the end with --dump), the wall time ATTRIBUTED to each kernel family: every instant is divided evenly between the
kernels running in it, so the column sums to the step's wall time minus idle (concurrent branches of the captured step
share the chip; a kernel's own duration says how long it was resident, not what it cost the step)."""
import argparse
import collections
import csv
import re


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([\w:]+(<[^>]*>)?)", n)
    return m.group(1)[:64] if m else n[:64]


def family(k):
    if "split_" in k:
        return "GEMM, bf16x3 split"
    if "_res_kernel" in k or "wgrad_first" in k:
        return "sa1 weight-resident (fp32)"
    if re.search(r"fps|ball_query|invert|three_nn|gather_rows", k):
        return "geometry"
    if "group" in k:
        return "grouping / affine"
    if re.search(r"gemm_|fewrow|bwd_pair|regw|ring_", k):
        return "GEMM, fp32 pipe"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--anchor", default="nll_fwd_kernel")
    ap.add_argument("--dump", type=int, default=0, help="print the kernel list of the N-th step from the end (1 = last full step)")
    ap.add_argument("--top", type=int, default=25)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    idx = [i for i, r in enumerate(rows) if a.anchor in r["Kernel_Name"]]
    if len(idx) < 3:
        raise SystemExit("anchor kernel %r launched %d times: need >= 3" % (a.anchor, len(idx)))
    steps = list(zip(idx[:-1], idx[1:]))

    def attribute(seg, t_end):
        ev = sorted(set([r["s"] for r in seg] + [r["e"] for r in seg] + [t_end]))
        attr, idle = collections.defaultdict(float), 0
        for x, y in zip(ev, ev[1:]):
            run = [r for r in seg if r["s"] <= x and r["e"] >= y]
            if not run:
                idle += y - x
            for r in run:
                attr[id(r)] += (y - x) / len(run)
        return attr, idle

    print("step   wall_us  kernels  idle_us  kernels per queue")
    for n, (i, j) in enumerate(steps[-8:]):
        seg = rows[i:j]
        _, idle = attribute(seg, rows[j]["s"])
        q = collections.Counter(r["Queue_Id"] for r in seg)
        print("%4d  %8.1f  %7d  %7.1f  %s" % (len(steps) - 8 + n, (rows[j]["s"] - rows[i]["s"]) / 1e3, len(seg), idle / 1e3,
                                            " ".join("q%s:%d" % kv for kv in sorted(q.items()))))
    for back in (2, 1):
        i, j = steps[-back]
        seg = rows[i:j]
        attr, idle = attribute(seg, rows[j]["s"])
        fam, agg = collections.defaultdict(float), collections.defaultdict(lambda: [0, 0.0, 0.0])
        for r in seg:
            k = short(r["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += attr[id(r)] / 1e3
            agg[k][2] += (r["e"] - r["s"]) / 1e3
            fam[family(k)] += attr[id(r)] / 1e3
        print("\nstep %d from the end: wall %.1f us, idle %.1f us; attributed us by family:" % (back, (rows[j]["s"] - rows[i]["s"]) / 1e3, idle / 1e3))
        for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
            print("  %8.1f  %s" % (v, k))
        if back == 1:
            print("  top kernels (attributed us | summed duration us | launches):")
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[: a.top]:
                print("  %8.1f  %8.1f  x%-2d %s" % (v[1], v[2], v[0], k))
    if a.dump:
        i, j = steps[-a.dump]
        t0 = rows[i]["s"]
        print("\nkernels of step %d from the end (start, end, duration us; queue; grid; name):" % a.dump)
        for r in rows[i:j]:
            print("%8.1f %8.1f %7.1f q%s g%-8s %s" % ((r["s"] - t0) / 1e3, (r["e"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, r["Queue_Id"],
                                                  r["Grid_Size_X"], short(r["Kernel_Name"])))


if __name__ == "__main__":
    main()
