"""numpy front-end of ``oracle/pn2_oracle.c`` (test infrastructure only, see ``oracle/__init__``)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("PN2_ORACLE_SO") or os.path.join(_HERE, "libpn2_oracle.so")     # (override: the sanitizer build, oracle/Makefile `asan`)


def build(force=False):
    """Compile the C restatement with the committed Makefile (gcc only)."""
    src = os.path.join(_HERE, "pn2_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libpn2_oracle.so"])
    host = os.path.join(_HERE, "libpn2_host.so")           # the same restatement behind the pn2_* symbols of include/pn2.h
    if force or not os.path.exists(host) or os.path.getmtime(host) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "pn2_host.c")),
                                                                           os.path.getmtime(os.path.join(_HERE, "..", "include", "pn2.h"))):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libpn2_host.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


_I = ctypes.c_int64


def square_distance(src, dst):
    """[B,S,3] x [B,N,3] -> [B,S,N]; bit form of pointnet_util.py:19-40."""
    src, dst = _f32(src), _f32(dst)
    B, S, _ = src.shape
    N = dst.shape[1]
    out = np.empty((B, S, N), np.float32)
    for b in range(B):
        lib().orc_square_distance(_p(src[b]), _p(dst[b]), _I(S), _I(N), _p(out[b]))
    return out


def farthest_point_sample(xyz, npoint, start):
    """[B,N,3], start[B] -> int64 [B,npoint]; pointnet_util.py:63-84 with the :75 draw passed in."""
    xyz, start = _f32(xyz), _i64(start)
    B, N, C = xyz.shape
    assert C == 3 and start.shape == (B,)
    out = np.empty((B, npoint), np.int64)
    lib().orc_fps(_p(xyz), _I(B), _I(N), _p(start), _I(npoint), _p(out))
    return out


def query_ball_point(radius, nsample, xyz, new_xyz):
    """pointnet_util.py:87-107.  Empty balls come back as N in every slot."""
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    if nsample > N:
        raise RuntimeError("nsample > N: the reference's slice/mask assignment fails here")
    out = np.empty((B, S, nsample), np.int64)
    r2 = np.float32(radius ** 2)
    lib().orc_ball_query(_p(xyz), _p(new_xyz), _I(B), _I(N), _I(S), ctypes.c_float(float(r2)), _I(nsample), _p(out))
    return out


def three_nn(xyz1, xyz2):
    """pointnet_util.py:295-297 -> (idx int64 [B,N,3], dist f32 [B,N,3]); ties to the lower index."""
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    assert S >= 3
    idx = np.empty((B, N, 3), np.int64)
    dist = np.empty((B, N, 3), np.float32)
    lib().orc_three_nn(_p(xyz1), _p(xyz2), _I(B), _I(N), _I(S), _p(idx), _p(dist))
    return idx, dist


def three_weights(dist):
    """pointnet_util.py:298-300."""
    dist = _f32(dist)
    w = np.empty_like(dist)
    lib().orc_three_weights(_p(dist), _I(dist.size // 3), _p(w))
    return w


def three_interpolate(points2, idx, w):
    """pointnet_util.py:301; points2 [B,S,D] -> [B,N,D]."""
    points2, idx, w = _f32(points2), _i64(idx), _f32(w)
    B, S, D = points2.shape
    N = idx.shape[1]
    out = np.empty((B, N, D), np.float32)
    lib().orc_three_interp(_p(points2), _p(idx), _p(w), _I(B), _I(N), _I(S), _I(D), _p(out))
    return out


def index_points(points, idx):
    """pointnet_util.py:43-60; idx [B,S] or [B,S,K]."""
    points, idx = _f32(points), _i64(idx)
    B, N, C = points.shape
    M = int(np.prod(idx.shape[1:]))
    out = np.empty((B, M, C), np.float32)
    rc = lib().orc_gather_rows(_p(points), _p(idx), _I(B), _I(N), _I(C), _I(M), _p(out))
    if rc != 0:
        raise IndexError("index out of range in index_points")
    return out.reshape(idx.shape + (C,))


def group(xyz, points, new_xyz, idx, xyz_first, ld=None):
    """pointnet_util.py:127-133 (xyz_first) / :243-251 (features first) -> [B,S,K,ld]."""
    xyz, new_xyz, idx = _f32(xyz), _f32(new_xyz), _i64(idx)
    B, N, _ = xyz.shape
    _, S, K = idx.shape
    D = 0 if points is None else points.shape[2]
    ld = 3 + D if ld is None else ld
    pts = None if points is None else _f32(points)
    out = np.empty((B, S, K, ld), np.float32)
    rc = lib().orc_group(_p(xyz), _p(pts) if pts is not None else None, _p(new_xyz), _p(idx),
                         _I(B), _I(N), _I(S), _I(K), _I(D), ctypes.c_int(1 if xyz_first else 0), _I(ld), _p(out))
    if rc != 0:
        raise IndexError("index out of range in group")
    return out
