"""The evidence tools that run on the development box: exercised on synthetic inputs (no GPU)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_step_timeline_attributes_wall_time_between_concurrent_kernels(tmp_path):
    """tools/step_timeline.py: steps are cut at the anchor kernel; an instant with k kernels running gives each 1/k of it, so the
    attributed times of a step add up to its wall time minus idle."""
    rows = ["Kind,Agent_Id,Queue_Id,Stream_Id,Thread_Id,Dispatch_Id,Kernel_Id,Kernel_Name,Correlation_Id,Start_Timestamp,End_Timestamp,"
            "LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Workgroup_Size_X,Workgroup_Size_Y,Workgroup_Size_Z,"
            "Grid_Size_X,Grid_Size_Y,Grid_Size_Z"]

    def k(q, name, s, e):
        rows.append('KERNEL_DISPATCH,Agent 2,%d,0,1,1,1,"%s",1,%d,%d,0,0,64,0,32,256,1,1,65536,1,1' % (q, name, s, e))
    for step in range(4):
        t0 = 10_000 * step
        k(1, "(anonymous namespace)::nll_fwd_kernel(float const*)", t0, t0 + 1000)
        k(1, "void (anonymous namespace)::split_nt_kernel<128, 4, 2, 1, 1, 0, false, 0, true, 1>(RegwArgs)", t0 + 1000, t0 + 5000)
        k(2, "void (anonymous namespace)::fps_kernel<512, 8, true>(float const*)", t0 + 3000, t0 + 7000)   # 2 us beside split_nt
        k(1, "void (anonymous namespace)::gemm_tn_kernel<64, 64, 32>(int)", t0 + 8000, t0 + 9000)          # 1 us idle before it
    trace = tmp_path / "t_kernel_trace.csv"
    trace.write_text("\n".join(rows) + "\n")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_timeline.py"), str(trace), "--dump", "1"],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    out = p.stdout
    assert "wall 10.0 us, idle 2.0 us" in out                     # 7 -> 8 us and 9 -> 10 us: nothing runs
    fam = {ln.split(None, 1)[1].strip(): float(ln.split()[0]) for ln in out.split("attributed us by family:")[-1].splitlines()[1:5]}
    assert abs(fam["GEMM, bf16x3 split"] - 3.0) < 1e-6            # 2 us alone + half of the 2 us shared with FPS
    assert abs(fam["geometry"] - 3.0) < 1e-6
    assert abs(fam["GEMM, fp32 pipe"] - 1.0) < 1e-6 and abs(fam["other"] - 1.0) < 1e-6
    assert abs(sum(fam.values()) - 8.0) < 1e-6                    # wall minus idle
