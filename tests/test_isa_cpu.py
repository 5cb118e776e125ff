"""CPU: checks on the GENERATED gfx950 code that the kernel sources rely on (hipcc cross-compiles without a GPU).

tools/check_isa.py: the hand-counted ``s_waitcnt vmcnt`` of the gather + conv kernel (csrc/grouped.hip) against the number of
vector-memory operations one loop trip really holds, its drained prologue, and "no spill" for the
register-stationary kernels.  The whole library's scratch table: ``python tools/check_isa.py scratch``.
"""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="no hipcc")
def test_hand_counted_waits_and_no_spills():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_isa.py"), "grouped", "scratch", "mlp_wide.hip", "grouped.hip"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="no hipcc")
def test_no_packed_fp32_operation_selects_the_high_half_of_a_vgpr_source():
    """Round 6: ``v_pk_*_f32 ... op_sel:[0,1]`` on a VGPR src1 is wrong in lanes 48..63 beside bf16 MFMAs on this hardware
    (csrc/pn2_common.h, PN2_OPAQUE: it cost pn2_fps its index-exactness in the captured step).  The built library must not hold one."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_isa.py"), "pkhi"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
