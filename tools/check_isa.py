#!/usr/bin/env python3
"""Build-time checks on the generated gfx950 code that the kernel sources RELY on (exit status 1 on any violation).

    python tools/check_isa.py            # all checks
    python tools/check_isa.py grouped    # the hand-counted waits of group_conv_fwd_kernel only
    python tools/check_isa.py scratch    # no kernel may spill beyond the allow-list below (private segment size)
    python tools/check_isa.py scratch mlp_wide.hip grouped.hip     # ... of these files only (what the CPU test suite runs)
    python tools/check_isa.py pkhi       # no packed-fp32 operation reads a VGPR pair through its HIGH half as second source

1. ``group_conv_fwd_kernel`` (csrc/grouped.hip) issues its gather as inline-asm loads and waits with HAND-COUNTED
   ``s_waitcnt vmcnt(kStores + 1)`` / ``vmcnt(kStores)`` (gfx9 retires loads and stores through one in-order counter; hipcc's
   own bookkeeping settles on vmcnt(0..2) there).  Those counts are only right while one trip of the slab loop holds exactly
   16 vector-memory loads and kStores = 3 + C_out / 4 stores, in the order  wait A, wait B, loads, stores, and while everything
   requested in the prologue has been waited for before the first trip (the prologue used to rely on dummy
   stores that the compiler removed).  A compiler upgrade that re-schedules the loop fails HERE, not silently on the GPU.
3. ``pkhi`` (round 6): on the MI355X of this pool ``v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32`` with ``op_sel:[0,1...]`` on a VGPR src1
   (the low result takes the HIGH dword of the pair) return wrong values in lanes 48..63 while another wave of the SIMD runs bf16
   MFMAs (csrc/pn2_common.h, PN2_OPAQUE; tools/exp/lds_reader_probe.hip).  hipcc emits the form by itself when it broadcasts the second
   register of a tuple; the sources avoid it and this check disassembles the BUILT library (every code object of its fat binary) to
   make sure a compiler change does not bring it back.
2. Scratch: a spilled register in a hand-scheduled kernel turns a counted wait into vmcnt(0) and costs far more than its
   28 bytes suggest (minor #12).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pointnet12_amd", "csrc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-DPN2_BUILD"]
FP_STRICT = {"geometry.hip", "train.hip"}           # built with -ffp-contract=off (csrc/Makefile)


def device_asm(src):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        flags = FLAGS + (["-ffp-contract=off"] if os.path.basename(src) in FP_STRICT else [])
        subprocess.run(["hipcc"] + flags + ["--cuda-device-only", "-S", src, "-o", out], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return open(out).read()


def kernels_of(asm):
    """{mangled name: [instruction lines]} of every kernel function in a device assembly listing."""
    out = {}
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)^\s*s_endpgm", asm, re.S | re.M):
        out[m.group(1)] = m.group(2).splitlines()
    return out


def check_grouped():
    errs = []
    ks = kernels_of(device_asm(os.path.join(CSRC, "grouped.hip")))
    found = 0
    for name, lines in ks.items():
        m = re.search(r"group_conv_fwd_kernelILi(\d+)E", name)
        if not m:
            continue
        found += 1
        co = int(m.group(1))
        k_stores = 3 + co // 4
        ops = []                                          # (kind, text) of everything that matters, in program order
        for ln in lines:
            t = ln.strip()
            if re.match(r"\.LBB\d+_\d+:", t):
                ops.append(("label", t.split(":")[0]))
            elif t.startswith("s_waitcnt") and "vmcnt" in t:
                ops.append(("wait", int(re.search(r"vmcnt\((\d+)\)", t).group(1))))
            elif re.match(r"(global|buffer|flat)_load", t):
                ops.append(("load", t))
            elif re.match(r"(global|buffer|flat)_store", t):
                ops.append(("store", t))
            elif re.match(r"(global|buffer|flat)_atomic", t):
                ops.append(("atomic", t))
            elif t.startswith("s_cbranch") or t.startswith("s_branch"):
                ops.append(("branch", t.split()[-1]))
        # the slab loop: the label whose body holds the hand-counted wait pair and ends with a branch back to it
        loop = None
        for i, (k, v) in enumerate(ops):
            if k != "label":
                continue
            for j in range(i + 1, len(ops)):
                if ops[j][0] == "label":
                    break
                if ops[j] == ("branch", v):
                    body = ops[i + 1:j]
                    if ("wait", k_stores + 1) in body and ("wait", k_stores) in body:
                        loop = (i, j, body)
                    break
        tag = "group_conv_fwd_kernel<%d>" % co
        if loop is None:
            errs.append("%s: no loop with s_waitcnt vmcnt(%d) and vmcnt(%d) found" % (tag, k_stores + 1, k_stores))
            continue
        i, j, body = loop
        kinds = [k for k, _ in body if k in ("load", "store", "atomic")]
        n_load, n_store = kinds.count("load"), kinds.count("store")
        if n_load != 16 or n_store != k_stores or "atomic" in kinds:
            errs.append("%s: one trip holds %d loads / %d stores (the counted waits need 16 / %d)" % (tag, n_load, n_store, k_stores))
        waits = [v for k, v in body if k == "wait"]
        if waits[:2] != [k_stores + 1, k_stores]:
            errs.append("%s: waits of one trip are %s (want vmcnt(%d) then vmcnt(%d) first)" % (tag, waits, k_stores + 1, k_stores))
        if any(v < k_stores for v in waits):
            errs.append("%s: a tighter wait than the hand-counted ones inside the loop (%s): the stores are being waited for" % (tag, waits))
        order = [k for k, _ in body if k in ("wait", "load", "store")]
        first_load, last_load = order.index("load"), len(order) - 1 - order[::-1].index("load")
        first_store = order.index("store")
        if not (order[:2] == ["wait", "wait"] and last_load < first_store and first_load > 1):
            errs.append("%s: program order inside the trip is not [wait A, wait B, loads, stores]" % tag)
        # prologue: the last vector-memory wait in front of the loop label must be vmcnt(0), with no load behind it
        pre = ops[:i]
        last_wait = max((n for n, (k, _) in enumerate(pre) if k == "wait"), default=None)
        if last_wait is None or pre[last_wait][1] != 0 or any(k in ("load", "store") for k, _ in pre[last_wait + 1:]):
            errs.append("%s: requests of the prologue are not drained (s_waitcnt vmcnt(0)) in front of the first trip" % tag)
    if found != 2:
        errs.append("expected the 32- and the 64-channel instantiation of group_conv_fwd_kernel, found %d" % found)
    return errs


# Kernels that are ALLOWED a private segment, with its ceiling in bytes per lane.  Round 5: the list holds only
# kernels that run in NONE of the five benchmark steps (checked against the rocprof kernel lists of profiles/r0*_prof_*):
#   * the register-resident FPS kernels at 24 .. 28 points per thread (N = 24 577 .. 28 672: a 1024-thread workgroup has 128
#     registers per lane, the cloud itself fills them; single-cloud inference sizes, no benchmark configuration),
#   * two streamed NT tiles that only serve layers the weight-resident kernels refuse (below 32 768 rows): 128 x 96 with a
#     BatchNorm loader (64 -> 96 forward) and 128 x 64 with the pooled dY loader -- two registers each.
# Gone in round 5: the pooled few-row data gradient (52 bytes; it runs in the MSG step: its launch bound capped the allocator at
# 256 registers where it needs 246 when left alone) and fwd_res_kernel<4, 4> (140 bytes; its launcher could never fit W plus eight
# staging buffers into the LDS -- dead code, no longer instantiated).
# Anything else -- in particular every register-stationary kernel of mlp_wide.hip, the fused backward of mlp_res.hip and the
# hand-counted gather + conv kernel -- must not spill at all.
ALLOWED_SCRATCH = [
    (r"^fps_pruned_kernel<1024, (24|25|26|28), ", 160),
    (r"^fps_rows_kernel<1024, (24|25|26|28), ", 112),
    (r"^gemm_nt_kernel<128, 96, 16, 4, 1, 3, 1, false, true, LoadBnRelu", 8),
    (r"^gemm_nt_kernel<128, 64, 32, 2, 2, 2, 1, true, true, LoadDyPooled, EpiDgradMask", 8),
]


def check_scratch(files=None):
    errs = []
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith(".hip") or (files and f not in files):
            continue
        flags = FLAGS + (["-ffp-contract=off"] if f in FP_STRICT else [])
        err = subprocess.run(["hipcc"] + flags + ["-c", os.path.join(CSRC, f), "-o", os.devnull,
                                                  "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
        cur = None
        for ln in err.splitlines():
            m = re.search(r"Function Name: (\S+)", ln)
            if m:
                cur = m.group(1)
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", ln)
            if m and cur and int(m.group(1)) != 0:
                name = subprocess.run(["c++filt", cur], capture_output=True, text=True).stdout.strip()
                name = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))
                cap = max([c for pat, c in ALLOWED_SCRATCH if re.search(pat, name)], default=0)
                if int(m.group(1)) > cap:
                    errs.append("%s: %s uses %s bytes of scratch per lane (allowed: %d)" % (f, name[:110], m.group(1), cap))
    return errs


LLVM_BIN = "/opt/rocm/lib/llvm/bin"


def check_pkhi(lib=None):
    """Every v_pk_{add,mul,fma}_f32 of the built library: src1 must not be a VGPR pair selected through its high half."""
    lib = lib or os.path.join(ROOT, "pointnet12_amd", "libpn2_hip.so")
    if not os.path.exists(lib):
        return ["pkhi: %s is not built" % lib]
    errs, total = [], 0
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
        data = open(fat, "rb").read()
        offs = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]       # one bundle per translation unit
        for i, o in enumerate(offs):
            part, co = os.path.join(d, "b%d.bin" % i), os.path.join(d, "b%d.co" % i)
            open(part, "wb").write(data[o:(offs[i + 1] if i + 1 < len(offs) else len(data))])
            subprocess.run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True, stderr=subprocess.DEVNULL)
            asm = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", co], capture_output=True, text=True).stdout
            cur = "?"
            for ln in asm.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
                if m:
                    cur = m.group(1)
                    continue
                m = re.search(r"\b(v_pk_(?:add|mul|fma)_f32)\s+(.*)", ln.split("//")[0])
                if not m:
                    continue
                total += 1
                sel = re.search(r"op_sel:\[[01],([01])", m.group(2))
                src1 = [f.strip() for f in m.group(2).split(",")][2]
                if sel and sel.group(1) == "1" and src1.startswith("v"):
                    name = subprocess.run(["c++filt", cur], capture_output=True, text=True).stdout.strip()
                    errs.append("pkhi: %s: %s %s" % (name[:100], m.group(1), m.group(2).strip()))
    if total < 1000:
        errs.append("pkhi: only %d packed fp32 operations found in %s -- the disassembly did not work" % (total, lib))
    return errs[:20]


def main():
    what = sys.argv[1:] or ["grouped", "scratch", "pkhi"]
    errs = []
    if "grouped" in what:
        errs += check_grouped()
    if "scratch" in what:
        errs += check_scratch([w for w in what if w.endswith(".hip")] or None)
    if "pkhi" in what:
        errs += check_pkhi()
    for e in errs:
        print("ISA CHECK FAILED: " + e)
    if not errs:
        print("isa checks ok (%s)" % ", ".join(what))
    return 1 if errs else 0


if __name__ == "__main__":
    sys.exit(main())
