#!/usr/bin/env python3
"""Per-kernel register / spill / occupancy table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py pointnet12_amd/csrc/mlp_res.hip [extra hipcc flags]

Exit status 1 when a kernel uses scratch beyond the allow-list of tools/check_isa.py (a spill in a hand-scheduled kernel).
"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-DPN2_BUILD", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for ln in out.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", ln)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_isa import ALLOWED_SCRATCH          # noqa: E402  (the documented spillers and their ceilings)
bad = 0
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["name"])
    name = re.sub(r"\(.*", "", name)
    cap = max([c for pat, c in ALLOWED_SCRATCH if re.search(pat, name.replace("void ", ""))], default=0)
    if r.get("ScratchSize", 0) > cap:
        bad += 1
    print("%-70s vgpr %3d agpr %3d spill %3d scratch %4d occ %d" % (name[:70], r.get("VGPRs", -1), r.get("AGPRs", -1),
          r.get("VGPRs Spill", -1), r.get("ScratchSize", -1), r.get("Occupancy", -1)))
if bad:
    sys.exit("%d kernel(s) spill beyond the allow-list of tools/check_isa.py" % bad)
