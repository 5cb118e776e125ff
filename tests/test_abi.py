"""CPU: the C-ABI library builds, loads and exports every symbol include/pn2.h declares."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from pointnet12_amd import _lib


def header_functions():
    text = open(os.path.join(ROOT, "include", "pn2.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pn2_[a-z0-9_]+)\s*\(", text)))


def test_header_matches_binding_table():
    assert header_functions() == sorted(_lib.SIGNATURES)


def test_library_exports_every_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_functions():
        assert hasattr(lib, name), name
    lib.pn2_version.restype = ctypes.c_int
    assert lib.pn2_version() == 4            # host-only call: no GPU needed
    lib.pn2_error_string.restype = ctypes.c_char_p
    assert lib.pn2_error_string(-1) == b"invalid argument"


def test_argument_checks_need_no_gpu():
    lib = _lib.load()
    assert lib.pn2_fps(None, 1, 1, None, 1, None, None, None) == -1
    lib.pn2_fps_workspace_bytes.restype = ctypes.c_int64
    assert lib.pn2_fps_workspace_bytes(2, 4096, 512) == 0
    assert lib.pn2_fps_workspace_bytes(2, 65536, 1024) == 2 * 1024 * 8 * 32     # 8 cooperating workgroups, 32-byte slots
    assert lib.pn2_fps_workspace_bytes(200, 65536, 1024) == 200 * 65536 * 4    # too many clouds to co-schedule: fallback


def test_cpu_tensors_are_refused():
    import torch
    from pointnet12_amd import pointnet_util as U
    with pytest.raises(_lib.Pn2Error):
        U.farthest_point_sample(torch.zeros(1, 8, 3), 2)
