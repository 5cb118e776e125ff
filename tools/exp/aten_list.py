#!/usr/bin/env python3
"""List the non-library (ATen / rocclr) kernels of ONE step of a rocprofv3 --kernel-trace CSV, each with the library kernels
around it: python tools/exp/aten_list.py trace.csv"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "nll_fwd_kernel" in r["Kernel_Name"]]
a, b = marks[-2], marks[-1]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n)
    return re.sub(r"\s+", " ", n)[:90]
mine = ("_kernel<", "_kernel(", "pn2_", "fps_", "gemm_", "regw_")
step = rows[a:b]
n_aten = 0
for i, r in enumerate(step):
    n = short(r["Kernel_Name"])
    aten = any(m in n for m in ("elementwise_kernel", "vectorized_elementwise", "reduce_kernel<", "CatArray", "softmax_warp", "fused_dropout", "rocclr", "masked_scale"))
    lib = not aten
    if not lib:
        n_aten += 1
        prev = short(step[i - 1]["Kernel_Name"])[:40] if i else ""
        nxt = short(step[i + 1]["Kernel_Name"])[:40] if i + 1 < len(step) else ""
        print("%4d %7.1f us grid %-9s wg %-5s %-80s | after %-40s | before %s" % (i, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
              r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"), n, prev, nxt))
print("kernels per step %d, non-library %d" % (len(step), n_aten))
