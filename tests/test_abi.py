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
    assert lib.pn2_version() == _lib.ABI_VERSION     # host-only call: no GPU needed
    lib.pn2_error_string.restype = ctypes.c_char_p
    assert lib.pn2_error_string(-1) == b"invalid argument"


def test_argument_checks_need_no_gpu():
    lib = _lib.load()
    assert lib.pn2_fps(None, 1, 1, None, 1, None, None, None) == -1
    lib.pn2_fps_workspace_bytes.restype = ctypes.c_int64
    assert lib.pn2_fps_workspace_bytes(2, 4096, 512) == 0
    assert lib.pn2_fps_workspace_bytes(2, 65536, 1024) == 2 * 1024 * 8 * 32     # 8 cooperating workgroups, 32-byte slots
    assert lib.pn2_fps_workspace_bytes(200, 65536, 1024) == 200 * 65536 * 4    # too many clouds to co-schedule: fallback


def test_options_are_explicit_and_the_library_never_reads_the_environment():
    """dispatch switches live in ONE table behind pn2_set_option / pn2_get_option (include/pn2.h); the library
    has no getenv of its own (the Python binding forwards PN2_* variables once, at load time)."""
    lib = _lib.load()
    opts = _lib.options()
    assert {"PN2_WIDE", "PN2_RING", "PN2_TN_SMALLP", "PN2_FPS_SINGLE_MAX", "PN2_BWD_PAIR"} <= set(opts)
    v = ctypes.c_int(-7)
    assert lib.pn2_get_option(b"PN2_TN_SMALLP", ctypes.addressof(v)) == 0 and v.value == opts["PN2_TN_SMALLP"]
    old = opts["PN2_NT_CFG"]
    try:
        assert lib.pn2_set_option(b"NT_CFG", 9) == 0                                  # with or without the prefix
        assert lib.pn2_get_option(b"PN2_NT_CFG", ctypes.addressof(v)) == 0 and v.value == 9
    finally:
        _lib.set_option("PN2_NT_CFG", old)
    assert lib.pn2_set_option(b"PN2_NO_SUCH_OPTION", 1) == -1 and lib.pn2_get_option(b"PN2_WIDE", None) == -1
    assert lib.pn2_option_name(len(opts)) is None and lib.pn2_option_name(-1) is None
    for f in os.listdir(_lib.CSRC):                                                  # no hidden switches
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(_lib.CSRC, f)).read(), f
    assert "thread_local" not in open(os.path.join(_lib.CSRC, "mlp.hip")).read()


def test_cpu_tensors_are_refused():
    import torch
    from pointnet12_amd import pointnet_util as U
    with pytest.raises(_lib.Pn2Error):
        U.farthest_point_sample(torch.zeros(1, 8, 3), 2)


def test_host_build_exports_the_geometry_symbols_and_matches_golden():
    """the same pn2_* symbols in a host (CPU) build of the restatement (oracle/pn2_host.c -> libpn2_host.so).
    Driven through the SAME ctypes signatures as the HIP library (host pointers), it reproduces the reference's golden
    FPS / ball-query indices and distance bits."""
    import numpy as np
    from conftest import golden
    from oracle import geometry as G
    G.build()
    host = ctypes.CDLL(os.path.join(ROOT, "oracle", "libpn2_host.so"))
    for name in ("pn2_version", "pn2_fps_workspace_bytes", "pn2_fps", "pn2_ball_query", "pn2_square_distance", "pn2_three_nn",
                 "pn2_gather_rows", "pn2_group", "pn2_three_interp"):
        fn = getattr(host, name)
        fn.restype, fn.argtypes = _lib.SIGNATURES[name]
    assert host.pn2_version() == _lib.ABI_VERSION
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    g = golden("g1_fps.npz")
    xyz, start, ref = g["kitti1024/xyz"], g["kitti1024/start"], g["kitti1024/idx"]
    out = np.empty(ref.shape, np.int64)
    assert host.pn2_fps(p(np.ascontiguousarray(xyz)), xyz.shape[0], xyz.shape[1], p(np.ascontiguousarray(start)), ref.shape[1], p(out),
                        None, None) == 0
    assert (out == ref).all()
    g2 = golden("g2_ball.npz")
    bx, bn, bref = np.ascontiguousarray(g2["kitti/xyz"]), np.ascontiguousarray(g2["kitti/new_xyz"]), g2["kitti/r0.2_k32"]
    bout = np.empty(bref.shape, np.int64)
    assert host.pn2_ball_query(p(bx), p(bn), bx.shape[0], bx.shape[1], bn.shape[1], float(np.float32(0.2 ** 2)), 32, p(bout), None) == 0
    assert (bout == bref).all()
    g3 = golden("g3_sqdist.npz")
    d = np.empty((1, g3["new_xyz"].shape[1], g3["xyz"].shape[1]), np.float32)
    assert host.pn2_square_distance(p(np.ascontiguousarray(g3["new_xyz"])), p(np.ascontiguousarray(g3["xyz"])), 1, d.shape[1], d.shape[2],
                                    p(d), None) == 0
    assert (d.view(np.uint32)[0, ::16, :] == g3["sample_bits"]).all()
