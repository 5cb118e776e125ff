"""MI355X-native counterpart of the reference's ``model/pointnet_util.py``.

Same public names, argument orders, tensor layouts and ``state_dict`` keys as the reference
(``PointNetSetAbstraction``, ``PointNetSetAbstractionMsg``, ``PointNetFeaturePropagation`` and
the six module-level functions), so ``model/pointnet2.py:5`` style imports work unchanged.
Underneath, every op is a hand-written gfx950 kernel reached through the C ABI of
``include/pn2.h`` (ctypes, raw device pointers, torch's current HIP stream).  PyTorch only
owns memory, streams and the autograd graph; there is no eager/CPU fallback -- a missing
``libpn2_hip.so`` or a non-GPU tensor raises.

Layout notes
  * function level: channel-last ``[B, N, C]`` (as the reference); module level: channel-first
    ``[B, C, N]``.  Modules return channel-first *views* of channel-last storage (the reference
    does the same for ``new_xyz``, pointnet_util.py:200), so the ``permute + contiguous`` at the
    top of the next module is free.
  * inside a module the grouped tensor is position-major ``[P, C]`` (P = B*S*K rows); the
    reference's ``[B, C, K, S]`` permute (pointnet_util.py:194) never happens.
"""
import contextlib
import ctypes
import weakref
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import check as _check, ptr as _p


def timeit(tag, t):                      # pointnet_util.py:7-9 (unused helper, kept for API parity)
    from time import time
    print("{}: {}s".format(tag, time() - t))
    return time()


def pc_normalize(pc):                    # pointnet_util.py:11-17 (unused helper, kept for API parity)
    pc = pc - np.mean(pc, axis=0)
    return pc / np.max(np.sqrt(np.sum(pc ** 2, axis=1)))


def _r4(c):
    return (c + 3) & ~3


def _gpu_f32(t, name):
    if not t.is_cuda:
        raise _lib.Pn2Error("%s must live on the GPU: this package has no CPU path" % name)
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be float32 (got %s)" % (name, t.dtype))   # the reference raises too (:74,82)
    return t.contiguous()


def _empty_rows(rows, cols, device):
    """Uninitialised [rows, round4(cols)] float32.  Every kernel that fills such a matrix writes the pad columns
    too (as zeros: GEMM epilogues, group, three_interp with zero_tail, bn_relu_max, relu_bwd_reduce), so no
    separate clearing pass is spent on it."""
    return torch.empty(rows, _r4(cols), device=device, dtype=torch.float32)


# Small zero-initialised scratch (BN statistics, reduction buffers, coefficient blocks, weight-gradient tiles) is
# carved from per-stream chunks that are cleared by ONE memset each, instead of one fill kernel per buffer: a step of
# MSG-SemSeg issued ~50 such fills, every one a ~5 us hop on the dependency chain of a stream (the fixed cost of a
# step is ~2 ms, measured by sweeping the batch size, and is made of exactly these hops).  A piece is handed out
# once; the chunk's storage is released when the last view dies.  Under stream capture the arena is only used
# inside a capture scope announced by pointnet12_amd.graph (a chunk allocated in one capture must not leak into
# another graph or into eager code: its memset is part of that graph only).
_ZERO_CHUNK_BYTES = 16 << 20
_zero_arenas = {}
_capture_scope = None


def set_capture_scope(token):
    """graph.GraphedStep brackets every stream capture with set_capture_scope(object()) / set_capture_scope(None)."""
    global _capture_scope
    _capture_scope = token
    for k in [k for k in _zero_arenas if k[2] is not None]:
        del _zero_arenas[k]


def _zeros_small(nbytes, device):
    """uint8[nbytes], zero, 256-byte aligned."""
    capturing = torch.cuda.is_current_stream_capturing()
    if nbytes > _ZERO_CHUNK_BYTES // 2 or (capturing and _capture_scope is None):
        return torch.zeros(nbytes, device=device, dtype=torch.uint8)
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, _capture_scope if capturing else None)
    a = _zero_arenas.get(key)
    if a is None or a[1] + nbytes > _ZERO_CHUNK_BYTES:
        a = _zero_arenas[key] = [torch.zeros(_ZERO_CHUNK_BYTES, device=device, dtype=torch.uint8), 0]
    off = a[1]
    a[1] = off + ((nbytes + 255) & ~255)
    return a[0][off:off + nbytes]


def _zeros_f32(shape, device):
    """A zero-initialised float32 tensor from the arena: the scatter targets of the backward pass (a few MB each) share the
    chunk's one clear instead of a fill launch apiece."""
    n = 1
    for d in shape:
        n *= int(d)
    return _zeros_small(4 * n, device).view(torch.float32).view(*shape)


_const_zeros = {}


def _zero_centres(B, C, device):
    """new_xyz of group_all (pointnet_util.py:151): a constant, created once per (device, B, C) outside any capture."""
    key = (device.index, B, C)
    t = _const_zeros.get(key)
    if t is None:
        t = torch.zeros(B, 1, C, device=device)
        if torch.cuda.is_current_stream_capturing():
            return t
        torch.cuda.current_stream(device).synchronize()
        _const_zeros[key] = t
    return t


def _contig_weight(w):
    """The [C_out, C_in(,1(,1))] Conv weight as the kernels read it: contiguous storage, row pitch C_in.  Parameters
    always are; a copy is made only for exotic views."""
    w = w.detach()
    return w if w.is_contiguous() else w.contiguous()


# Weights whose rows are not 16-byte aligned (C_in = 137 of fp1's first layer, 323, 515 ...) take the GEMM kernels' guarded
# scalar loaders (fwd 65 536 x 137 -> 128: 60 TF; its dgrad: 37 TF).  From this many rows on, a zero-padded [C_out, round4(C_in)]
# copy is made once per forward by pn2_copy_cols (one ~5 us launch) and the forward and the data-gradient GEMM read that.
ALIGN_WEIGHT_MIN_ROWS = int(os.environ.get("PN2_ALIGN_WEIGHT_MIN_ROWS", "32768"))


def _aligned_weight(w, ci, co, P):
    """(tensor to read, row pitch): ``w`` itself, or its zero-padded copy when C_in % 4 != 0 and the layer is long enough."""
    wc = _contig_weight(w)
    if ci % 4 == 0 or P < ALIGN_WEIGHT_MIN_ROWS:
        return wc, ci
    ld = _r4(ci)
    wp = _zeros_f32((co, ld), wc.device)
    _check(_lib.load().pn2_copy_cols(_p(wc), ci, 0, _p(wp), ld, 0, co, ci, _lib.stream()), "pn2_copy_cols")
    return wp, ld


def _padded_feature_columns(w, D, ci, xyz_first):
    """The factorised first layer reads the feature columns straight out of the [C_out, 3+D] weight (pitch 3+D, K = D).  With
    the features FIRST and D % 4 == 1 that slice is 16-byte aligned with a pitch of round4(D): the GEMMs would take their
    float4 path, whose contract is ZERO pad entries (include/pn2.h) -- but the "pad" quad here holds the xyz weights (ADVICE
    round 3: the forward then leans on X's pad columns being exactly 0, the data gradient writes non-zero pad lanes).  For
    exactly that case a zero-padded [C_out, round4(D)] copy is made (one pn2_copy_cols launch); every other layout either
    satisfies the contract or takes the guarded scalar loaders."""
    if xyz_first or D % 4 == 0 or ci % 4 != 0:
        return None
    wc = _contig_weight(w)
    co = wc.shape[0]
    ld = _r4(D)
    wp = _zeros_f32((co, ld), wc.device)
    _check(_lib.load().pn2_copy_cols(_p(wc), ci, 0, _p(wp), ld, 0, co, D, _lib.stream()), "pn2_copy_cols")
    return wp


_ident_cache = {}
_bcast_cache = {}


def _broadcast_neighbours(B, N, device, want_inv):
    """(idx int64 [B,N,3] = 0, weight float32 [B,N,3] = (1, 0, 0), members, owners): FeaturePropagation below a group_all
    stage (S == 1) as an interpolation.  members / owners: the target-sorted form of idx (pn2_invert_index) for the segmented
    backward -- with every entry pointing at row 0 the plain scatter would queue 3 N atomics per channel on one address.
    Constants, created once per (device, B, N) outside any capture."""
    key = (device.index, B, N)
    t = _bcast_cache.get(key)
    if t is None or (want_inv and t[2] is None):
        idx = torch.zeros(B, N, 3, device=device, dtype=torch.int64)
        w = torch.zeros(B, N, 3, device=device, dtype=torch.float32)
        w[:, :, 0] = 1.0
        members = owners = None
        if want_inv:
            members = torch.empty(B, 3 * N, device=device, dtype=torch.int32)
            owners = torch.empty(B, 3 * N, device=device, dtype=torch.int32)
            scratch = torch.empty(B, 3, device=device, dtype=torch.int32)
            _check(_lib.load().pn2_invert_index(_p(idx), B, 3 * N, 1, _p(members), _p(owners), _p(scratch), _lib.stream()),
                   "pn2_invert_index")
        t = (idx, w, members, owners)
        if torch.cuda.is_current_stream_capturing():
            return t                                   # private to this capture; not cached
        torch.cuda.current_stream(device).synchronize()
        _bcast_cache[key] = t
    return t


def _ident_coef(co, device):
    """BN-backward coefficient block of the identity, dY := 1*dZ + 0*(y - 0) + 0 (float[4*round4(co)]); constant,
    created once per (device, co) outside any capture."""
    key = (device.index, co)
    t = _ident_cache.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            t = torch.zeros(4 * _r4(co), device=device, dtype=torch.float32)
            t[:co] = 1.0
            return t                                   # private to this capture; not cached
        t = torch.zeros(4 * _r4(co), device=device, dtype=torch.float32)
        t[:co] = 1.0
        torch.cuda.current_stream(device).synchronize()    # once per (device, co): later readers may be on any stream
        _ident_cache[key] = t
    return t


# --------------------------------------------------------------------------------------- primitives

def square_distance(src, dst):
    """[B,N,3] x [B,M,3] -> [B,N,M]; bit form of pointnet_util.py:19-40."""
    src, dst = _gpu_f32(src, "src"), _gpu_f32(dst, "dst")
    B, N, _ = src.shape
    M = dst.shape[1]
    out = torch.empty(B, N, M, device=src.device, dtype=torch.float32)
    _check(_lib.load().pn2_square_distance(_p(src), _p(dst), B, N, M, _p(out), _lib.stream()), "pn2_square_distance")
    return out


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx, checked):
        B, N, C = points.shape
        M = idx.numel() // B
        out = torch.empty(B, M, C, device=points.device, dtype=torch.float32)
        err = _zeros_small(4, points.device).view(torch.int32) if checked else None        # (the zero arena: no fill launch)
        _check(_lib.load().pn2_gather_rows(_p(points), _p(idx), B, N, C, M, _p(out), _p(err), _lib.stream()),
               "pn2_gather_rows")
        # the reference's advanced indexing raises on the host; here that costs one device->host read, which a stream
        # capture cannot contain: under capture the check is skipped (the kernel clamps nothing -- callers that capture
        # pass indices they produced themselves: FPS / ball query / 3-NN outputs are in range by construction)
        if checked and not torch.cuda.is_current_stream_capturing() and int(err.item()) != 0:
            raise IndexError("index out of range in index_points")     # as the reference's advanced indexing
        ctx.save_for_backward(idx)
        ctx.shape = (B, N, C, M)
        return out.view(tuple(idx.shape) + (C,))

    @staticmethod
    def backward(ctx, grad):
        (idx,) = ctx.saved_tensors
        B, N, C, M = ctx.shape
        grad = grad.contiguous()
        gp = _zeros_f32((B, N, C), grad.device)
        _check(_lib.load().pn2_gather_rows_bwd(_p(grad), _p(idx), B, N, C, M, _p(gp), _lib.stream()),
               "pn2_gather_rows_bwd")
        return gp, None, None


def index_points(points, idx, _checked=True):
    """points [B,N,C], idx [B,S] or [B,S,K] (int64) -> [B,S,C] / [B,S,K,C]; pointnet_util.py:43-60."""
    points = _gpu_f32(points, "points")
    idx = idx.contiguous()
    if idx.dtype != torch.int64:
        idx = idx.long()
    return _GatherRows.apply(points, idx, _checked)


MSG_SCALE_STREAMS = os.environ.get("PN2_MSG_STREAMS", "1") == "1"     # measured: 10.36 -> 9.90 ms/step (MSG-SemSeg)
MSG_LAST_SCALE_ON_MAIN = os.environ.get("PN2_MSG_MAIN_LAST", "1") == "1"
_scale_stream_pool = {}


def _scale_streams(device, n):
    pool = _scale_stream_pool.setdefault(str(device), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool


_fps_start_feed = None


def set_fps_start_feed(feed):
    """Route the FPS start draws through ``feed.take(B, N, device)`` (pointnet12_amd.graph) or back to eager (None)."""
    global _fps_start_feed
    _fps_start_feed = feed


def draw_fps_start(B, N, device):
    """The start draw of pointnet_util.py:75: one CPU-generator randint per FPS call, then H2D.

    While a hipGraph of the step is being captured the draw is deferred: the feed hands out a slice of a device
    buffer that is refilled before every replay from the SAME CPU generator, in the same call order."""
    if _fps_start_feed is not None:
        return _fps_start_feed.take(B, N, device)
    return torch.randint(0, N, (B,), dtype=torch.long).to(device, non_blocking=True)


class GeometryTape:
    """Recorded results of the index-producing primitives (FPS, ball query, 3-NN) of one forward pass.

    Everything those primitives compute depends only on the xyz coordinates, never on features or weights, so the
    whole geometry of a batch can be produced ahead of time (pointnet12_amd.graph prefetches the NEXT batch's
    geometry on a side stream while the current batch's MLPs run).  ``record`` mode: the modules run their
    geometry, append the results here and skip all feature work (they return uninitialised placeholders of the
    right shape).  ``replay`` mode: the primitives hand back the recorded tensors in the same call order."""

    def __init__(self, into=None):
        self.items = []
        self.pos = 0
        self.mode = "record"
        self.into = into           # record mode: existing items (same call order / shapes) to overwrite instead

    def rewind(self, mode):
        self.pos = 0
        self.mode = mode


_tape = None


def set_geometry_tape(tape):
    global _tape
    _tape = tape


def get_geometry_tape():
    return _tape


def _recording():
    return _tape is not None and _tape.mode == "record"


def _dest(*specs):
    """Output tensors of the primitive being recorded, one per (shape, dtype) in ``specs``.  Recording ``into`` another tape's
    items (the second graph of pointnet12_amd.graph's pair writes where the first one reads) the primitive is handed THOSE
    tensors and its kernel writes them directly -- the first version computed into fresh tensors and copied: 23 extra launches
    in every second step of MSG-SemSeg (rocprofv3 kernel trace: 161 against 138 kernels, +0.3 ms)."""
    dst = None
    if _tape is not None and _tape.mode == "record" and _tape.into is not None:
        dst = _tape.into[len(_tape.items)]
        dst = dst if isinstance(dst, tuple) else (dst,)
        dev = torch.device("cuda", torch.cuda.current_device())
        if len(dst) != len(specs) or any(tuple(d.shape) != tuple(sh) or d.dtype != dt or not d.is_contiguous() or d.device != dev
                                         for d, (sh, dt) in zip(dst, specs)):
            dst = None                            # (a different call order / shape: _taped copies, as before)
    return dst


def _taped(compute):
    if _tape is None:
        return compute()
    if _tape.mode == "record":
        v = compute()
        if _tape.into is not None:            # land the result in the tensors another graph replays from
            dst = _tape.into[len(_tape.items)]
            for d, x in zip(dst if isinstance(dst, tuple) else (dst,), v if isinstance(v, tuple) else (v,)):
                if d is not x:                # (not already written in place: see _dest)
                    d.copy_(x)
            v = dst
        _tape.items.append(v)
        return v
    v = _tape.items[_tape.pos]
    _tape.pos += 1
    return v


def _placeholder(B, C, S, device):
    """Channel-first view of uninitialised channel-last storage (what a module returns in record mode)."""
    return torch.empty(B, S, C, device=device, dtype=torch.float32).permute(0, 2, 1)


def farthest_point_sample(xyz, npoint, start=None):
    """xyz [B,N,3] -> int64 [B,npoint]; pointnet_util.py:63-84.  ``start`` overrides the random draw."""
    if _tape is not None and _tape.mode == "replay":
        return _taped(None)
    xyz = _gpu_f32(xyz, "xyz")
    B, N, C = xyz.shape
    if C != 3:
        raise RuntimeError("farthest_point_sample needs 3 coordinates (pointnet_util.py:79)")
    if start is None:
        start = draw_fps_start(B, N, xyz.device)
    else:
        start = torch.as_tensor(start)
        if not start.is_cuda and (start.numel() != B or int(start.min()) < 0 or int(start.max()) >= N):
            raise IndexError("farthest_point_sample: start must hold B indices in [0, N)")   # free on host tensors;
            # device tensors are not read back (that would synchronise): the kernels clamp them into the cloud
    start = start.to(device=xyz.device, dtype=torch.int64).contiguous()
    dst = _dest(((B, npoint), torch.int64))
    out = dst[0] if dst else torch.empty(B, npoint, device=xyz.device, dtype=torch.int64)
    lib = _lib.load()
    nbytes = lib.pn2_fps_workspace_bytes(B, N, npoint)
    work = torch.empty(nbytes, device=xyz.device, dtype=torch.uint8) if nbytes else None
    _check(lib.pn2_fps(_p(xyz), B, N, _p(start), npoint, _p(out), _p(work), _lib.stream()), "pn2_fps")
    return _taped(lambda: out)


def _sampled_centres(xyz, fps_idx):
    """new_xyz = index_points(xyz, fps_idx) (pointnet_util.py:125, :238) as part of the GEOMETRY: it depends on the coordinates
    alone (xyz never carries a gradient), so under a GeometryTape it is recorded with the indices -- the prefetch
    branch gathers the next batch's centres, and the step's own chain is one launch per sampling stage shorter."""
    if _tape is not None and _tape.mode == "replay":
        return _taped(None)
    B, N, C = xyz.shape
    dst = _dest((tuple(fps_idx.shape) + (C,), torch.float32))
    if dst and fps_idx.dtype == torch.int64 and fps_idx.is_contiguous():
        xyz = _gpu_f32(xyz, "xyz")                # (index_points' launch, written where the other graph reads)
        _check(_lib.load().pn2_gather_rows(_p(xyz), _p(fps_idx), B, N, C, fps_idx.numel() // B, _p(dst[0]), None, _lib.stream()),
               "pn2_gather_rows")
        return _taped(lambda: dst[0])
    return _taped(lambda: index_points(xyz, fps_idx, _checked=False).detach())


def query_ball_point(radius, nsample, xyz, new_xyz):
    """-> int64 [B,S,nsample]; pointnet_util.py:87-107 (first nsample in-radius indices, padded with the first)."""
    if _tape is not None and _tape.mode == "replay":
        return _taped(None)
    xyz, new_xyz = _gpu_f32(xyz, "xyz"), _gpu_f32(new_xyz, "new_xyz")
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    if nsample > N:
        raise RuntimeError("nsample (%d) > N (%d): the reference's mask assignment fails here too" % (nsample, N))
    dst = _dest(((B, S, nsample), torch.int64))
    out = dst[0] if dst else torch.empty(B, S, nsample, device=xyz.device, dtype=torch.int64)
    r2 = float(np.float32(radius ** 2))
    lib = _lib.load()
    wb = lib.pn2_ball_query_workspace_bytes(B, N, S)          # > 0: the library wants to take the centres in spatial order
    work = torch.empty(wb, device=xyz.device, dtype=torch.uint8) if wb else None
    _check(lib.pn2_ball_query_ws(_p(xyz), _p(new_xyz), B, N, S, r2, nsample, _p(out), _p(work), _lib.stream()), "pn2_ball_query_ws")
    return _taped(lambda: out)


def three_nn(xyz1, xyz2):
    """3 nearest of xyz2 [B,S,3] for every point of xyz1 [B,N,3] -> (idx int64, raw dist, weights), each [B,N,3].

    pointnet_util.py:295-300; ties go to the lower index (the reference's sort leaves tie order undefined)."""
    if _tape is not None and _tape.mode == "replay":
        return _taped(None)
    xyz1, xyz2 = _gpu_f32(xyz1, "xyz1"), _gpu_f32(xyz2, "xyz2")
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    if S < 3:
        raise RuntimeError("three_nn needs S >= 3 (S == 2 raises in the reference as well)")
    dst = _dest(((B, N, 3), torch.int64), ((B, N, 3), torch.float32), ((B, N, 3), torch.float32))
    if dst:
        idx, dist, w = dst
    else:
        idx = torch.empty(B, N, 3, device=xyz1.device, dtype=torch.int64)
        dist = torch.empty(B, N, 3, device=xyz1.device, dtype=torch.float32)
        w = torch.empty(B, N, 3, device=xyz1.device, dtype=torch.float32)
    _check(_lib.load().pn2_three_nn(_p(xyz1), _p(xyz2), B, N, S, _p(idx), _p(dist), _p(w), _lib.stream()),
           "pn2_three_nn")
    return _taped(lambda: (idx, dist, w))


class _Group(torch.autograd.Function):
    """Gather K neighbours per centroid, centre xyz, concat features -> position-major [B*S*K, ld]."""

    @staticmethod
    def forward(ctx, xyz, points, new_xyz, idx, S, K, xyz_first):
        B, N, _ = xyz.shape
        D = 0 if points is None else points.shape[2]
        rows = _empty_rows(B * S * K, 3 + D, xyz.device)
        ld = rows.shape[1]
        _check(_lib.load().pn2_group(_p(xyz), _p(points), _p(new_xyz), _p(idx), B, N, S, K, D, int(xyz_first), ld,
                                     _p(rows), None, _lib.stream()), "pn2_group")
        ctx.save_for_backward(idx)
        ctx.dims = (B, N, S, K, D, int(xyz_first), ld)
        return rows

    @staticmethod
    def backward(ctx, grad_rows):
        (idx,) = ctx.saved_tensors
        B, N, S, K, D, xyz_first, ld = ctx.dims
        gp = None
        if D > 0 and ctx.needs_input_grad[1]:
            grad_rows = grad_rows.contiguous()
            gp = _zeros_f32((B, N, D), grad_rows.device)
            _check(_lib.load().pn2_group_bwd(_p(grad_rows), _p(idx), B, N, S, K, D, xyz_first, ld, _p(gp),
                                             _lib.stream()), "pn2_group_bwd")
        return None, gp, None, None, None, None, None


# The scatter-adds of the backward pass (3-NN interpolation, factorised first layer) run as segmented reductions over
# the neighbour index SORTED BY TARGET (csrc/scatter.hip): ~10x fewer atomics, perfectly balanced.  The index comes
# from the geometry alone, so the sort is taped with it and runs on the prefetch stream, one step ahead.
GATHER_BACKWARD = os.environ.get("PN2_GATHER_BWD", "1") == "1"


def _inverse_index(idx2d, T):
    """idx2d [B, M] int64 with values in [0, T) -> (members int32 [B, M], owners int32 [B, M]): the positions of each
    cloud sorted by the value they point at, and that value."""
    def compute():
        B, M = idx2d.shape
        dev = idx2d.device
        dst = _dest(((B, M), torch.int32), ((B, M), torch.int32))
        members = dst[0] if dst else torch.empty(B, M, device=dev, dtype=torch.int32)
        owners = dst[1] if dst else torch.empty(B, M, device=dev, dtype=torch.int32)
        scratch = torch.empty(B, 2 * T + 1, device=dev, dtype=torch.int32)
        _check(_lib.load().pn2_invert_index(_p(idx2d), B, M, T, _p(members), _p(owners), _p(scratch), _lib.stream()),
               "pn2_invert_index")
        return (members, owners)
    return _taped(compute)


class _InterpCat(torch.autograd.Function):
    """cat([points1, three_interpolate(points2)], -1) written straight into one [B*N, ld] matrix."""

    @staticmethod
    def forward(ctx, points1, points2, idx, w, inv_off=None, inv_mem=None):
        B, S, D2 = points2.shape
        N = idx.shape[1]
        D1 = 0 if points1 is None else points1.shape[2]
        rows = _empty_rows(B * N, D1 + D2, points2.device)
        ld = rows.shape[1]
        lib, st = _lib.load(), _lib.stream()
        p1 = points1.contiguous() if D1 else None       # one launch: copy of points1 + interpolation behind it
        _check(lib.pn2_three_interp(_p(points2), _p(idx), _p(w), B, N, S, D2, _p(rows), ld, D1, 1, _p(p1), st),
               "pn2_three_interp")
        ctx.save_for_backward(idx, w, inv_off, inv_mem)
        ctx.dims = (B, N, S, D1, D2, ld)
        return rows

    @staticmethod
    def backward(ctx, grad_rows):
        idx, w, inv_off, inv_mem = ctx.saved_tensors
        B, N, S, D1, D2, ld = ctx.dims
        grad_rows = grad_rows.contiguous()
        lib, st = _lib.load(), _lib.stream()
        g1 = g2 = None
        if D1 and ctx.needs_input_grad[0]:
            g1 = grad_rows.view(B, N, ld)[:, :, :D1]        # a strided view: no copy kernel
        if ctx.needs_input_grad[1]:
            g2 = _zeros_f32((B, S, D2), grad_rows.device)
        if ctx.needs_input_grad[1] and inv_off is not None:         # segmented reduction over the target-sorted index
            _check(lib.pn2_three_interp_bwd_seg(_p(grad_rows), ld, D1, _p(inv_off), _p(inv_mem), _p(w), B, N, S, D2, _p(g2), st),
                   "pn2_three_interp_bwd_seg")
        elif ctx.needs_input_grad[1]:
            _check(lib.pn2_three_interp_bwd(_p(grad_rows), ld, D1, _p(idx), _p(w), B, N, S, D2, _p(g2), st),
                   "pn2_three_interp_bwd")
        return g1, g2, None, None, None, None


# --------------------------------------------------------------------------------------- shared MLP

# the factorised first layer's dWx partials go through replicated scratch copies (pn2_group_affine_bwd_seg); 0 = straight
# atomics on the weight gradient, for A/B runs
DWX_REPLICAS_SCRATCH = os.environ.get("PN2_DWX_REPLICAS", "1") != "0"
_DIRECT_GRADS = False
# Statistics -> affine block / backward coefficients inside the kernel that finishes the reduction (the "tails" of
# include/pn2.h) instead of pn2_bn_finalize / pn2_bn_bwd_coef launches.  Measured on MI355X: no gain (MSG-SemSeg
# B=16 9.257 vs 9.245 ms, B=1 2.29 vs 2.33 ms; SSG 3.833 vs 3.827 ms) -- the last workgroup's ticket + device-scope
# reads cost the same ~5 us as the dependent launch they replace -- so the stand-alone launches stay the default.
FUSED_BN_TAILS = os.environ.get("PN2_FUSED_BN_TAILS", "0") == "1"
# Consumer-side BatchNorm (round 4, ABI 8): the statistics -> affine block step and the reductions -> coefficients step run as a
# prologue of the first kernel that READS the block (every workgroup recomputes it from the producer's finished fp64 sums) instead
# of pn2_bn_finalize / pn2_bn_bwd_coef launches of their own -- 50 hops of ~5 us on the dependency chain of an MSG-SemSeg step, 44
# of SSG's.  Same fp64 arithmetic per channel: bit-identical blocks.  0: the stand-alone launches (A/B runs).
LAZY_BN = os.environ.get("PN2_LAZY_BN", "1") == "1"
# dgrad + wgrad of a layer as one call / one launch on the few-row and mid-size layers (pn2_conv1x1_bwd_pair); 0: A/B runs
BWD_PAIR = os.environ.get("PN2_BWD_PAIR_CALL", "1") == "1"
# first-layer weight gradients without a data gradient read dZ and the input rows only (closed-form BatchNorm-backward terms);
# 0: the general kernel, which reads Y as well (A/B runs)
WGRAD_CF = os.environ.get("PN2_WGRAD_CF", "1") != "0"
# the last layer of a pooled MLP records the per-group extrema in its GEMM epilogue (pn2_conv1x1_fwd_pool); 0: A/B runs
POOL_IN_EPILOGUE = os.environ.get("PN2_POOL_EPILOGUE", "1") == "1"
# narrow first layers: gather + first conv in one launch (pn2_group_conv_fwd); 0: pn2_group then the GEMM (A/B runs)
GATHER_CONV = os.environ.get("PN2_GATHER_CONV", "1") == "1"
# MSG scales write their pooled outputs into column slices of one matrix instead of torch.cat; 0: A/B runs
MSG_CONCAT_IN_PLACE = os.environ.get("PN2_MSG_CONCAT_IN_PLACE", "1") == "1"
# Few-row layers (sa3 / fp3 / fp2 of the segmentation nets: P = 2 k .. 8 k rows) put ONE workgroup on a CU and keep its matrix
# pipe ~30 % busy (csrc/mlp.hip, dispatch_nt_vec); their weight-gradient GEMM feeds nothing downstream, so it is issued on a
# companion stream and shares the CUs with the data-gradient GEMM of the same layer.  0 = off (A/B); rows up to which it is done.
WGRAD_SIDE_MAX_ROWS = int(os.environ.get("PN2_WGRAD_SIDE_MAX_ROWS", "0"))
_wgrad_streams = {}


def _wgrad_side_stream(device):
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    s = _wgrad_streams.get(key)
    if s is None:
        s = _wgrad_streams[key] = torch.cuda.Stream(device=device)
    return s


_REPL = 8                         # PN2_STAT_REPLICAS of include/pn2.h
_PAIR_RUNS_SPLIT = set()          # (P, C_out, C_in, pooled, masked) shapes pn2_conv1x1_bwd_pair answered with PN2_OK_SPLIT
_MALL_CHUNK_BYTES = 1 << 62       # row-chunked dgrad+wgrad pairing is OFF: measured 10.9 -> 13.1 ms/step at 96 MiB chunks
                                  # (per-launch fixed costs beat the Infinity-Cache hits); kept as a tuning knob


def set_direct_grad_accumulation(enabled):
    """When enabled, the MLP backward adds weight / BatchNorm gradients straight into existing ``param.grad`` tensors
    (the wgrad kernel's atomics and the BN-coefficient kernel write there) and returns None for them, instead of
    materialising per-layer gradient tensors that autograd then adds with ~100 tiny kernels per step.
    ``parallel.FlatGradBucket(..., direct=True)`` turns this on: every ``.grad`` is then a slice of one flat buffer
    that is zeroed once per step.  Parameters without a pre-existing contiguous fp32 ``.grad`` fall back to autograd."""
    global _DIRECT_GRADS
    _DIRECT_GRADS = bool(enabled)


def _direct_ok(params):
    if not _DIRECT_GRADS:
        return False
    for q in params:
        g = q.grad
        if g is None or not g.is_contiguous() or g.dtype != torch.float32 or g.device != q.device:
            return False
    return True


class _SharedMLP(torch.autograd.Function):
    """L x (1x1 conv + BatchNorm + ReLU) on position-major rows, then max over ``pool`` consecutive rows.

    forward(rows [P, ld0], c_in, pool, training, bn_cfg, geom, *flat) with ``flat`` = per layer
    (weight, bias, gamma, beta, running_mean, running_var, num_batches_tracked) and
    ``bn_cfg`` = per layer (eps, momentum).  pool == 0: no pooling, output [P, C_L] (FP);
    pool == K: output [P/K, C_L] (SA).  Pre-BN activations of every layer are kept for backward.

    ``dest`` = None, or (buffer [G, ld], first column): the pooled output is written straight into that column slice of a wider
    matrix (the concatenated output of the MSG scales, pointnet_util.py:260) instead of a tensor of its own that a
    ``torch.cat`` would copy; the returned tensor is that slice.

    ``geom`` = None, or (xyz [B,N,3], new_xyz [B,S,3], idx [B,S,K], xyz_first) for the FACTORISED first layer:
    ``rows`` is then the un-grouped feature tensor [B,N,D] and layer 1 is evaluated as
    ``Zf[b, idx] + W_x (xyz[idx] - centre)`` with ``Zf = W_f f + bias`` computed once per source point
    (csrc/grouped.hip) -- the grouped [P, 3+D] tensor never exists and the K-fold redundant GEMM rows vanish.
    """

    @staticmethod
    def forward(ctx, rows, c_in, pool, training, bn_cfg, geom, dest, *flat):
        lib, st = _lib.load(), _lib.stream()
        dev = rows.device
        L = len(flat) // 7
        if geom is None:
            P = rows.shape[0]
        else:
            g_xyz, g_new, g_idx, g_first, g_inv = geom[:5]
            gB, gN, gD = rows.shape
            gS, gK = g_idx.shape[1], g_idx.shape[2]
            P = gB * gS * gK
        chans = [c_in] + [flat[7 * l].shape[0] for l in range(L)]
        n_stats = _REPL * 2 * sum(chans[1:]) if training else 0
        n_aff = 4 * sum(_r4(c) for c in chans[1:])
        zero_bytes = _zeros_small(8 * n_stats + 4 * n_aff + 4 * L, dev)
        stats = zero_bytes[:8 * n_stats].view(torch.float64) if training else None
        Ys, affs = [], []
        aff_all = zero_bytes[8 * n_stats:8 * n_stats + 4 * n_aff].view(torch.float32)
        tickets = zero_bytes.data_ptr() + 8 * n_stats + 4 * n_aff      # one uint32 per fused BatchNorm tail
        aff_off = 0
        x, ldx, x_aff, off = rows, rows.shape[-1], None, 0
        pool_ws = None
        gather = geom is not None and len(geom) > 5 and bool(geom[5])      # (xyz, new_xyz, idx, xyz_first, None, True)
        gathered = None
        w_pads = {}
        lazy_on = training and LAZY_BN and not FUSED_BN_TAILS
        lazy_prev = None                   # pn2_bn_lazy of the previous layer's block: realised by the next launch that reads it
        keep = []                          # (ctypes structures must outlive the calls that take their address)

        def in_lazy():
            return None if lazy_prev is None else ctypes.byref(lazy_prev)
        for l in range(L):
            w, b, gamma, beta, rmean, rvar, nbt = flat[7 * l:7 * l + 7]
            co, ci = chans[l + 1], chans[l]
            pooled_last = (l == L - 1 and pool and training and x_aff is not None and POOL_IN_EPILOGUE and P % 32 == 0 and
                           (pool == 16 or pool % 32 == 0) and co % 32 == 0 and not FUSED_BN_TAILS)
            # the pooled last layer of the long sa1 stacks: its pre-BN output is never written -- the backward runs on the layer's
            # input (pn2_conv1x1_bwd_cf) -- where the library has that form (csrc/mlp_res.hip: split_bwd_cf_kernel)
            no_y = bool(pooled_last and lazy_on and lib.pn2_conv1x1_bwd_cf_supported(P, co, ci, pool))
            y = None if no_y else _empty_rows(P, co, dev)
            st_l = stats[off:off + _REPL * 2 * co] if training else None
            aff = aff_all[aff_off:aff_off + 4 * _r4(co)]
            aff_off += 4 * _r4(co)
            eps, mom = bn_cfg[l]
            fin = None
            if training and FUSED_BN_TAILS:     # the producer of the statistics also turns them into the affine block
                fin = ctypes.byref(_lib.BnFinalizeTail(tickets + 4 * l, _p(gamma), _p(beta), eps, mom, _p(rmean), _p(rvar),
                                                       _p(nbt), _p(aff)))
            if l == 0 and gather:
                # narrow first layer: gather + conv in one launch; the grouped rows are written once, for the backward
                gathered = _empty_rows(P, ci, dev)
                _check(lib.pn2_group_conv_fwd(_p(g_xyz), _p(rows), _p(g_new), _p(g_idx), gB, gN, gS, gK, gD, int(g_first),
                                              _p(_contig_weight(w)), ci, _p(b), _p(gathered), gathered.shape[1], _p(y), y.shape[1], co,
                                              _p(st_l), st), "pn2_group_conv_fwd")
            elif l == 0 and geom is not None:
                # the kernels read the xyz / feature columns straight out of the [co, 3+D] weight (pitch ci): no copies
                w_ptr = _contig_weight(w).data_ptr()
                wx_ptr, wf_ptr = (w_ptr, w_ptr + 12) if g_first else (w_ptr + 4 * gD, w_ptr)
                wf_ld = ci
                wf_pad = _padded_feature_columns(w, gD, ci, g_first)
                if wf_pad is not None:                  # (D % 4 == 1, features first: see _padded_feature_columns)
                    w_pads["wf"] = wf_pad
                    wf_ptr, wf_ld = wf_pad.data_ptr(), wf_pad.shape[1]
                ldd = _r4(gD)
                feat = rows.reshape(gB * gN, gD)
                if ldd != gD:
                    feat = torch.nn.functional.pad(feat, (0, ldd - gD))
                zf = _empty_rows(gB * gN, co, dev)
                _check(lib.pn2_conv1x1_fwd(_p(feat), ldd, None, wf_ptr, wf_ld, _p(b), _p(zf), zf.shape[1], gB * gN, gD, co,
                                           None, None, None, st), "pn2_conv1x1_fwd")
                _check(lib.pn2_group_affine_fwd(_p(zf), zf.shape[1], _p(g_xyz), _p(g_new), _p(g_idx), wx_ptr, ci, gB, gN, gS,
                                                gK, co, _p(y), y.shape[1], _p(st_l), fin, st), "pn2_group_affine_fwd")
            elif pooled_last and fin is None:
                # last layer of a pooled MLP: the weight-resident kernel also records the per-group extrema of y, so the
                # pooled output needs no second pass over Y (unsupported shapes: the plain launch + pn2_bn_relu_max below)
                pool_ws = torch.empty(2 * (P // pool) * co, device=dev, dtype=torch.float32)
                rc = lib.pn2_conv1x1_fwd_pool(_p(x), ldx, _p(x_aff), _p(_contig_weight(w)), ci, _p(b), _p(y), co, P, ci, co,
                                              _p(st_l), pool, _p(gamma), _p(pool_ws), in_lazy(), st)
                if rc == _lib.PN2_EUNSUPPORTED and y is None:      # (no output-free form for this shape after all)
                    y = _empty_rows(P, co, dev)
                    rc = lib.pn2_conv1x1_fwd_pool(_p(x), ldx, _p(x_aff), _p(_contig_weight(w)), ci, _p(b), _p(y), co, P, ci, co,
                                                  _p(st_l), pool, _p(gamma), _p(pool_ws), in_lazy(), st)
                if rc == _lib.PN2_EUNSUPPORTED:
                    pool_ws = None
                    _check(lib.pn2_conv1x1_fwd(_p(x), ldx, _p(x_aff), _p(_contig_weight(w)), ci, _p(b), _p(y), y.shape[1], P, ci,
                                               co, _p(st_l), fin, in_lazy(), st), "pn2_conv1x1_fwd")
                else:
                    _check(rc, "pn2_conv1x1_fwd_pool")
            else:
                w_rd, w_ld = _aligned_weight(w, ci, co, P)
                if w_ld != ci:
                    w_pads[l] = w_rd                            # the data-gradient GEMM of the backward reads it too
                _check(lib.pn2_conv1x1_fwd(_p(x), ldx, _p(x_aff), _p(w_rd), w_ld, _p(b), _p(y), y.shape[1], P, ci,
                                           co, _p(st_l), fin, in_lazy(), st), "pn2_conv1x1_fwd")
            lazy_prev = None                                    # (whatever block the launch above read has been filled)
            if fin is None and lazy_on:
                # this layer's statistics become its affine block inside the next launch that reads the block
                lazy_prev = _lib.BnLazy(_p(st_l), _p(gamma), _p(beta), eps, mom, _p(rmean), _p(rvar), _p(nbt), _p(aff), P, co)
                keep.append(lazy_prev)
            elif fin is None:
                _check(lib.pn2_bn_finalize(_p(st_l), P, co, _p(gamma), _p(beta), eps, mom, int(training),
                                           _p(rmean), _p(rvar), _p(nbt), _p(aff), st), "pn2_bn_finalize")
            Ys.append(y)
            affs.append(aff)
            x, ldx, x_aff = y, (y.shape[1] if y is not None else 0), aff
            off += _REPL * 2 * co
        cl = chans[-1]
        K = pool if pool else 1
        G = P // K
        if dest is not None and pool and cl % 4 == 0:
            big, col0 = dest
            out = big[:, col0:col0 + cl]                       # pitch of the wide matrix; the kernels take it as ldo
        else:
            out = _empty_rows(G, cl, dev)
        ldo = out.stride(0)
        # the argmax shares the output's pitch inside the kernels (arg + g * ldo + c): its own buffer, same pitch
        arg = torch.empty(G, ldo, device=dev, dtype=torch.int32) if pool else None
        if pool_ws is not None:
            _check(lib.pn2_bn_pool_select(_p(pool_ws), _p(affs[-1]), G, cl, _p(out), ldo, _p(arg), in_lazy(), st), "pn2_bn_pool_select")
        else:
            _check(lib.pn2_bn_relu_max(_p(Ys[-1]), Ys[-1].shape[1], _p(affs[-1]), G, K, cl, _p(out), ldo, _p(arg), in_lazy(), st),
                   "pn2_bn_relu_max")
        if training:
            bump_param_generation()             # running statistics were written through raw pointers
        ctx.meta = (chans, pool, bool(training), P)
        ctx.geom = None if geom is None else (g_xyz, g_new, g_idx, bool(g_first), g_inv)   # index/coordinate tensors: no cycle
        ctx.gather = (gB, gN, gD) if gather else None
        if gather:
            rows = gathered                     # what the backward reads as the first layer's input
        ctx.params = flat                       # leaf parameters / buffers (no grad_fn): no cycle either
        ctx.w_pads = w_pads                     # plain scratch tensors (no grad_fn)
        # save_for_backward (not ctx attributes): `out` is this node's own output, and holding it on ctx would close a
        # reference cycle that only the cyclic GC breaks -- gigabytes of saved activations would pile up for several
        # steps and the caching allocator would stall in hipMalloc/hipFree in the middle of a step.
        # (Ys[-1] is None where the last layer ran without an output; its backward needs the extrema records instead)
        ctx.save_for_backward(rows, out, arg, *Ys, *affs, *[flat[7 * l] for l in range(L)],
                              *[flat[7 * l + 2] for l in range(L)], pool_ws if Ys[-1] is None else None)
        return out[:, :cl] if out.shape[1] != cl else out

    @staticmethod
    def backward(ctx, grad_out):
        lib, st = _lib.load(), _lib.stream()
        chans, pool, training, P = ctx.meta
        saved = ctx.saved_tensors
        L = len(chans) - 1
        rows, out, arg = saved[0], saved[1], saved[2]
        Ys, affs = saved[3:3 + L], saved[3 + L:3 + 2 * L]
        Ws, gammas = saved[3 + 2 * L:3 + 3 * L], saved[3 + 3 * L:3 + 4 * L]
        dev = rows.device
        cl = chans[-1]
        ldo = out.stride(0)                    # (a column slice of a wider matrix when the forward was given a ``dest``)
        # The pooled branch reads the incoming gradient once, scalar-wise, at any pitch: a column slice of a wider matrix (a
        # branch of a concatenated MSG output) or an un-padded [G, cl] matrix is passed in place -- no copy kernel at the head
        # of the branch's backward chain.
        ld_grad = ldo
        if pool and grad_out.dim() == 2 and grad_out.stride(1) == 1 and grad_out.stride(0) >= cl and grad_out.dtype == torch.float32:
            ld_grad = grad_out.stride(0)
        else:
            if ldo != cl:
                g = torch.zeros(out.shape[0], ldo, device=dev, dtype=torch.float32)
                g[:, :cl] = grad_out
                grad_out = g
            grad_out = grad_out.contiguous()
        n_red = _REPL * 2 * sum(chans[1:])
        direct = _direct_ok([ctx.params[7 * l + j] for l in range(L) for j in (0, 2, 3)])
        sizes = [4 * _r4(chans[l + 1]) + (0 if direct else chans[l + 1] * chans[l] + chans[l + 1]) for l in range(L)]
        zero_bytes = _zeros_small(8 * n_red + 4 * sum(sizes) + 4 * L, dev)
        red = zero_bytes[:8 * n_red].view(torch.float64)
        tickets = zero_bytes.data_ptr() + 8 * n_red + 4 * sum(sizes)   # one uint32 per fused BatchNorm tail
        offs = np.cumsum([0] + [_REPL * 2 * c for c in chans[1:]])
        K = pool if pool else 1
        G = P // K
        flat = ctx.params
        # zeroed scratch: BN-backward coefficient blocks (+ the gradient tensors in autograd mode)
        zbuf = zero_bytes[8 * n_red:8 * n_red + 4 * sum(sizes)].view(torch.float32)
        zoff = np.cumsum([0] + sizes)
        outs = []                                    # per layer: coef, dgamma, dbeta, dW, dbias
        for l in range(L):
            co, ci = chans[l + 1], chans[l]
            z0 = int(zoff[l])
            coef = zbuf[z0:z0 + 4 * _r4(co)]
            if direct:
                outs.append((coef, flat[7 * l + 2].grad, flat[7 * l + 3].grad, flat[7 * l].grad, None))
            else:
                outs.append((coef, torch.empty(co, device=dev, dtype=torch.float32),
                             torch.empty(co, device=dev, dtype=torch.float32),
                             zbuf[z0 + 4 * _r4(co):z0 + 4 * _r4(co) + co * ci].view(co, ci),
                             zbuf[z0 + 4 * _r4(co) + co * ci:z0 + 4 * _r4(co) + co * ci + co]))
        coef_done = [False] * L

        def coef_tail(l):
            """The producer of layer l's reductions also turns them into coef_l / dgamma_l / dbeta_l (fused tail)."""
            if not FUSED_BN_TAILS:
                return None
            coef_done[l] = True
            return ctypes.byref(_lib.BnCoefTail(tickets + 4 * l, _p(gammas[l]), _p(affs[l]), int(training), _p(outs[l][0]),
                                                _p(outs[l][1]), _p(outs[l][2]), int(direct)))
        dZ = None
        red_L = red[offs[L - 1]:offs[L]]
        dzp = None
        no_y = pool and Ys[-1] is None                 # the last layer ran without an output: its backward from the layer's input
        if no_y:
            pool_ws = saved[3 + 4 * L]
            dzp = torch.empty(G, ldo, device=dev, dtype=torch.float32)
            _check(lib.pn2_pool_bwd_reduce_rec(_p(grad_out), ld_grad, _p(out), ldo, _p(arg), _p(pool_ws), _p(affs[-1]), G, K, cl, _p(dzp),
                                               _p(red_L), _p(_contig_weight(Ws[-1])), chans[-2], _p(flat[7 * (L - 1) + 1]), _p(Ys[-2]),
                                               Ys[-2].shape[1], _p(affs[-2]), chans[-2], st), "pn2_pool_bwd_reduce_rec")
        elif pool:
            dzp = torch.empty(G, ldo, device=dev, dtype=torch.float32)    # dOut masked by out > 0: the pooled form of dZ_L the GEMM loaders read (pitch ldo, as arg)
            _check(lib.pn2_pool_bwd_reduce_ld(_p(grad_out), ld_grad, _p(out), ldo, _p(arg), _p(Ys[-1]), Ys[-1].shape[1], _p(affs[-1]),
                                              G, K, cl, _p(dzp), _p(red_L), coef_tail(L - 1), st), "pn2_pool_bwd_reduce_ld")
        else:
            dZ = _empty_rows(P, cl, dev)
            _check(lib.pn2_relu_bwd_reduce(_p(grad_out), ldo, _p(out), _p(Ys[-1]), Ys[-1].shape[1], _p(affs[-1]), P, cl,
                                           _p(dZ), dZ.shape[1], _p(red_L), coef_tail(L - 1), st), "pn2_relu_bwd_reduce")
        grads = [None] * (7 * L)
        d_rows = None
        # sa1-style stacks (a first layer on the 48-byte grouped rows, nobody needs d rows): the second layer's fused backward forms
        # the first layer's sum dZ^T x from its dX tiles (pn2_conv1x1_bwd_first) -- dZ1 is never written -- and the first layer's
        # weight gradient finishes from the input's moments (pn2_conv1x1_wgrad_cf with dZ == NULL)
        fuse_first = bool(L >= 3 and ctx.gather is not None and not ctx.needs_input_grad[0] and training and WGRAD_CF and LAZY_BN and
                          not FUSED_BN_TAILS and chans[0] <= 12 and chans[1] % 16 == 0 and chans[1] <= 128 and rows.shape[1] % 4 == 0 and
                          rows.shape[1] >= 12 and Ys[1] is not None and
                          lib.pn2_conv1x1_bwd_first_supported(P, chans[2], chans[1], chans[0]))
        first_scratch = None
        side, side_used = None, False
        if 0 < P <= WGRAD_SIDE_MAX_ROWS:
            main_stream = torch.cuda.current_stream(dev)
            side = _wgrad_side_stream(dev)
        for l in range(L - 1, -1, -1):
            co, ci = chans[l + 1], chans[l]
            y, aff = Ys[l], affs[l]
            ldy = y.shape[1] if y is not None else 0
            coef, dgamma, dbeta, dW, dbias = outs[l]
            w_p = flat[7 * l]
            # consumer-side BatchNorm backward: the first launch below that reads `coef` fills it from the reductions
            coef_lazy = None
            first_layer_special = l == 0 and ctx.geom is not None and ctx.gather is None
            # (the factorised first layer: its segmented scatter kernel is the consumer; the atomics form is not)
            seg_ok = not first_layer_special or (ctx.geom[4] is not None and co <= 256)
            if not coef_done[l] and training and LAZY_BN and seg_ok:
                cl_struct = _lib.BnCoefLazy(_p(red[offs[l]:offs[l + 1]]), _p(gammas[l]), _p(aff), _p(coef), _p(dgamma), _p(dbeta),
                                            int(direct), P, co)
                coef_lazy = ctypes.byref(cl_struct)
            elif not coef_done[l]:
                _check(lib.pn2_bn_bwd_coef(_p(red[offs[l]:offs[l + 1]]), P, co, _p(gammas[l]), _p(aff), int(training),
                                           _p(coef), _p(dgamma), _p(dbeta), int(direct), st), "pn2_bn_bwd_coef")
            if not direct:
                grads[7 * l + 1], grads[7 * l + 2], grads[7 * l + 3] = dbias, dgamma, dbeta
            if l == 0 and ctx.geom is not None and ctx.gather is None:
                d_rows, dW0 = _SharedMLP._first_layer_bwd(ctx, rows, Ws[0], dZ, y, coef, co, ctx.needs_input_grad[0],
                                                          w_p.grad if direct else None, coef_lazy)
                grads[0] = dW0
                continue
            x = rows if l == 0 else Ys[l - 1]
            x_aff = None if l == 0 else affs[l - 1]
            ldx = x.shape[1]
            pooled = dZ is None
            if not training and direct:
                raise NotImplementedError("direct gradient accumulation with eval-mode BatchNorm: use autograd mode")
            need_dx = l > 0 or ctx.needs_input_grad[0]
            if l == 1 and fuse_first and dZ is not None:
                first_scratch = _zeros_small(int(lib.pn2_conv1x1_wgrad_cf_scratch_bytes()), dev)
                rc = lib.pn2_conv1x1_bwd_first(_p(dZ), dZ.shape[1], _p(y), ldy, _p(coef), _p(_contig_weight(Ws[l])), ci, _p(x), ldx, _p(x_aff),
                                               _p(red[offs[0]:offs[1]]), _p(dW), ci, _p(rows), rows.shape[1], chans[0], _p(first_scratch),
                                               P, co, ci, coef_lazy, st)
                if rc != _lib.PN2_EUNSUPPORTED:
                    _check(rc, "pn2_conv1x1_bwd_first")
                    if not direct:
                        grads[7 * l] = dW.view_as(Ws[l])
                    dZ = None                                       # (the first layer's dZ exists only as the sums above)
                    continue
                first_scratch = None
            if l == 0 and first_scratch is not None:
                _check(lib.pn2_conv1x1_wgrad_cf(None, 0, _p(coef), _p(x), ldx, _p(_contig_weight(Ws[l])), ci, _p(flat[7 * l + 1]),
                                                _p(first_scratch), _p(dW), ci, P, co, ci, coef_lazy, st), "pn2_conv1x1_wgrad_cf")
                if not direct:
                    grads[7 * l] = dW.view_as(Ws[l])
                continue
            if y is None:
                # pooled last layer without its output: dX and dW from (dZp, arg) and the layer's input alone
                scratch = torch.empty(int(lib.pn2_conv1x1_bwd_cf_scratch_bytes(co, ci)), device=dev, dtype=torch.uint8)
                dx = _empty_rows(P, ci, dev)
                _check(lib.pn2_conv1x1_bwd_cf(_p(dzp), ldo, _p(arg), K, _p(coef), _p(_contig_weight(Ws[l])), ci, _p(flat[7 * l + 1]), _p(x), ldx,
                                              _p(x_aff), _p(dx), dx.shape[1], _p(red[offs[l - 1]:offs[l]]), _p(dW), ci, P, co, ci, coef_lazy,
                                              _p(scratch), st), "pn2_conv1x1_bwd_cf")
                if not direct:
                    grads[7 * l] = dW.view_as(Ws[l])
                dZ = dx
                continue
            if (need_dx and training and not FUSED_BN_TAILS and
                    lib.pn2_bwd_res_supported(P, co, ci, K if pooled else 0, int(x_aff is not None))):
                # narrow, long layer: dgrad + wgrad in ONE pass over dZ / Y / Y_prev, weights resident in LDS (mlp_res.hip)
                dx = _empty_rows(P, ci, dev) if l > 0 else torch.empty(P, ldx, device=dev, dtype=torch.float32)
                if l == 0:
                    d_rows = dx
                c_dz = (None, 0) if pooled else (_p(dZ), dZ.shape[1])
                c_pool = (_p(dzp), ldo, _p(arg), K) if pooled else (None, 0, None, 0)
                _check(lib.pn2_conv1x1_bwd(*c_dz, *c_pool, _p(y), ldy, _p(coef), _p(_contig_weight(Ws[l])), ci, _p(x), ldx,
                                           _p(x_aff), _p(dx), dx.shape[1], _p(red[offs[l - 1]:offs[l]]) if l > 0 else None,
                                           _p(dW), ci, P, co, ci, coef_lazy, st), "pn2_conv1x1_bwd")
                if not direct:
                    grads[7 * l] = dW.view_as(Ws[l])
                if l > 0:
                    dZ = dx
                continue
            dx = None
            if need_dx:
                w_l = _p(_contig_weight(Ws[l]))             # [co, ci] as stored: the dgrad kernel reads it down the columns
                w_ld = ci
                if l in ctx.w_pads:                         # its 16-byte aligned, zero-padded copy made by the forward
                    w_l, w_ld = _p(ctx.w_pads[l]), ctx.w_pads[l].shape[1]
                if l > 0:
                    dx = _empty_rows(P, ci, dev)
                else:
                    dx = d_rows = torch.empty(P, ldx, device=dev, dtype=torch.float32)   # pad lanes written (0) by the GEMM
            # dgrad and wgrad of one layer read the same dZ / Y rows.  Run them back to back on row chunks small
            # enough to stay in the 256 MiB Infinity Cache, so the second kernel's operand stream is served on-die
            # instead of from HBM (both only accumulate: fp64 reductions / fp32 weight gradients).
            row_bytes = 4 * ((1 if pooled else 2) * ldy + 2 * ldx)
            chunk = P
            if need_dx and P * row_bytes > _MALL_CHUNK_BYTES:
                chunk = max(_MALL_CHUNK_BYTES // row_bytes, 1 << 14)
                chunk -= chunk % (K * 128)                    # whole pooling groups, whole row tiles
                chunk = max(chunk, K * 128)
            for r0 in range(0, P, chunk):
                rn = min(chunk, P - r0)
                c_dz = (None, 0) if pooled else (dZ.data_ptr() + 4 * r0 * dZ.shape[1], dZ.shape[1])
                c_pool = (dzp.data_ptr() + 4 * (r0 // K) * ldo, ldo, arg.data_ptr() + 4 * (r0 // K) * ldo, K) if pooled \
                    else (None, 0, None, 0)
                c_y = y.data_ptr() + 4 * r0 * ldy
                c_x = x.data_ptr() + 4 * r0 * ldx
                pair_key = (P, co, ci, pooled, l > 0)
                if (need_dx and BWD_PAIR and training and chunk == P and side is None and not FUSED_BN_TAILS and
                        pair_key not in _PAIR_RUNS_SPLIT):
                    # data gradient and weight gradient of this layer as ONE call: on the few-row / mid-size layers both kernel
                    # bodies share one launch (pn2_conv1x1_bwd_pair); elsewhere the library issues the two launches itself
                    prev = (c_x, ldx, _p(x_aff), dx.data_ptr(), dx.shape[1], _p(red[offs[l - 1]:offs[l]])) if l > 0 else \
                        (None, 0, None, dx.data_ptr(), ldx, None)
                    rc = lib.pn2_conv1x1_bwd_pair(*c_dz, *c_pool, c_y, ldy, _p(coef), w_l, w_ld, *prev, c_x, ldx, _p(x_aff), _p(dW), ci,
                                                  rn, co, ci, coef_lazy, st)
                    if rc != _lib.PN2_OK_SPLIT:                 # (1: done as two launches -- not an error)
                        _check(rc, "pn2_conv1x1_bwd_pair")
                    else:
                        # the library ran this shape as pn2_conv1x1_dgrad + pn2_conv1x1_wgrad: from now on those two calls are made
                        # here (same launches, same results) -- a per-launch accounting (bench.py) then sees each kernel by itself
                        _PAIR_RUNS_SPLIT.add(pair_key)
                    coef_lazy = None
                    continue
                if need_dx:
                    c_dx = dx.data_ptr() + 4 * r0 * dx.shape[1]
                    if l > 0:
                        _check(lib.pn2_conv1x1_dgrad(*c_dz, *c_pool, c_y, ldy, _p(coef), w_l, w_ld, c_x, ldx, _p(x_aff), c_dx,
                                                     dx.shape[1], _p(red[offs[l - 1]:offs[l]]), rn, co, ci,
                                                     coef_tail(l - 1) if chunk == P else None, coef_lazy, st), "pn2_conv1x1_dgrad")
                    else:
                        _check(lib.pn2_conv1x1_dgrad(*c_dz, *c_pool, c_y, ldy, _p(coef), w_l, w_ld, None, 0, None, c_dx,
                                                     ldx, None, rn, co, ci, None, coef_lazy, st), "pn2_conv1x1_dgrad")
                    coef_lazy = None                            # filled: the weight gradient below reads it
                if (WGRAD_CF and l == 0 and training and not need_dx and not pooled and ci <= 15 and co % 16 == 0 and co <= 128
                        and chunk == P):
                    # first layer without a data gradient: the BatchNorm-backward terms of dY in closed form from the input rows'
                    # first and second moments -- the pass reads dZ and the rows, not Y (pn2_conv1x1_wgrad_cf, include/pn2.h)
                    scratch = _zeros_small(int(lib.pn2_conv1x1_wgrad_cf_scratch_bytes()), dev)
                    _check(lib.pn2_conv1x1_wgrad_cf(c_dz[0], c_dz[1], _p(coef), c_x, ldx, _p(_contig_weight(Ws[l])), ci,
                                                    _p(flat[7 * l + 1]), _p(scratch), _p(dW), ci, rn, co, ci, coef_lazy, st),
                           "pn2_conv1x1_wgrad_cf")
                    coef_lazy = None
                    continue
                if side is not None and chunk == P:
                    # the weight gradient on the companion stream: ordered behind everything issued so far on this stream (the
                    # coefficients, dZ), joined once at the end of this backward
                    side.wait_stream(main_stream)
                    with torch.cuda.stream(side):
                        _check(lib.pn2_conv1x1_wgrad(*c_dz, *c_pool, c_y, ldy, _p(coef), c_x, ldx, _p(x_aff), _p(dW), ci,
                                                     None if training else _p(dbias), rn, co, ci, coef_lazy, _lib.stream()), "pn2_conv1x1_wgrad")
                    side_used = True
                else:
                    ws_bytes = lib.pn2_conv1x1_wgrad_workspace_bytes(rn, co, ci, int(pooled)) if training else 0
                    if ws_bytes:                                # two-phase dW flush of the full-tile kernel: caller scratch
                        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
                        _check(lib.pn2_conv1x1_wgrad_ws(*c_dz, *c_pool, c_y, ldy, _p(coef), c_x, ldx, _p(x_aff), _p(dW), ci, None, rn, co, ci,
                                                        coef_lazy, _p(ws), st), "pn2_conv1x1_wgrad_ws")
                    else:
                        _check(lib.pn2_conv1x1_wgrad(*c_dz, *c_pool, c_y, ldy, _p(coef), c_x, ldx, _p(x_aff), _p(dW), ci,
                                                     None if training else _p(dbias), rn, co, ci, coef_lazy, st), "pn2_conv1x1_wgrad")
                coef_lazy = None
            if not direct:
                grads[7 * l] = dW.view_as(Ws[l])
            if l > 0:
                dZ = dx
        if side_used:
            main_stream.wait_stream(side)
        if ctx.gather is not None and d_rows is not None:          # gradient of the grouped rows -> the gathered source points
            gB, gN, gD = ctx.gather
            g_idx = ctx.geom[2]
            gp = _zeros_f32((gB, gN, gD), dev)
            _check(lib.pn2_group_bwd(_p(d_rows), _p(g_idx), gB, gN, g_idx.shape[1], g_idx.shape[2], gD, int(ctx.geom[3]),
                                     d_rows.shape[1], _p(gp), st), "pn2_group_bwd")
            d_rows = gp
        return (d_rows, None, None, None, None, None, None) + tuple(grads)

    @staticmethod
    def _first_layer_bwd(ctx, feats, w, dZ, y, coef, co, need_dfeat, w_grad, coef_lazy=None):
        """Backward of the factorised first layer: scatter dY to the source points, then two small GEMMs.
        ``w_grad``: None (return dW) or the [co, 3+D] gradient tensor to accumulate into (returns None)."""
        lib, st = _lib.load(), _lib.stream()
        g_xyz, g_new, g_idx, g_first, g_inv = ctx.geom
        B, N, D = feats.shape
        S, K = g_idx.shape[1], g_idx.shape[2]
        dev = feats.device
        ldc, ldd = _r4(co), _r4(D)
        ident = _ident_coef(co, dev)                  # dY := 1*G + 0*(y-0) + 0
        dW = _zeros_small(4 * co * (3 + D), dev).view(torch.float32).view(co, 3 + D) if w_grad is None else w_grad
        ldw = 3 + D
        x_col, f_col = (0, 3) if g_first else (D, 0)
        G = _zeros_f32((B * N, ldc), dev)
        if g_inv is not None and co <= 256:           # segmented reduction over the source-sorted ball-query index
            scratch = _zeros_small(4 * _lib.DWX_REPLICAS * 3 * ldc, dev) if DWX_REPLICAS_SCRATCH else None
            _check(lib.pn2_group_affine_bwd_seg(_p(dZ), dZ.shape[1], _p(y), y.shape[1], _p(coef), _p(g_xyz), _p(g_new),
                                                _p(g_inv[0]), _p(g_inv[1]), B, N, S, K, co, _p(G), ldc,
                                                dW.data_ptr() + 4 * x_col, ldw, _p(scratch), coef_lazy, st), "pn2_group_affine_bwd_seg")
        else:
            _check(lib.pn2_group_affine_bwd(_p(dZ), dZ.shape[1], _p(y), y.shape[1], _p(coef), _p(g_xyz), _p(g_new),
                                            _p(g_idx), B, N, S, K, co, _p(G), ldc, dW.data_ptr() + 4 * x_col, ldw, st),
                   "pn2_group_affine_bwd")
        feat = feats.reshape(B * N, D)
        if ldd != D:
            feat = torch.nn.functional.pad(feat, (0, ldd - D)).contiguous()
        _check(lib.pn2_conv1x1_wgrad(_p(G), ldc, None, 0, None, 0, _p(G), ldc, _p(ident), _p(feat), ldd, None,
                                     dW.data_ptr() + 4 * f_col, ldw, None, B * N, co, D, None, st), "pn2_conv1x1_wgrad")
        d_feats = None
        if need_dfeat:
            wf_ptr = _contig_weight(w).data_ptr() + (12 if g_first else 0)      # feature columns of the [co, 3+D] weight
            wf_ld = 3 + D
            if "wf" in ctx.w_pads:                   # their zero-padded copy made by the forward (D % 4 == 1, features first)
                wf_ptr, wf_ld = ctx.w_pads["wf"].data_ptr(), ctx.w_pads["wf"].shape[1]
            dF = _empty_rows(B * N, D, dev)
            _check(lib.pn2_conv1x1_dgrad(_p(G), ldc, None, 0, None, 0, _p(G), ldc, _p(ident), wf_ptr, wf_ld, None, 0, None,
                                         _p(dF), ldd, None, B * N, co, D, None, None, st), "pn2_conv1x1_dgrad")
            d_feats = (dF[:, :D] if ldd != D else dF).reshape(B, N, D)
        return d_feats, (dW.view_as(w) if w_grad is None else None)


class _Conv1x1(torch.autograd.Function):
    """Plain per-point linear layer on position-major rows (no BatchNorm): y = x W^T + b.

    The 128 -> num_classes classifier of the segmentation heads (reference model/pointnet2.py:174); a skinny GEMM
    (N = 13) that the vendor library runs at 0.11 ms -- the same fp32-MFMA NT/TN cores do it in ~15 us.
    Backward reuses pn2_conv1x1_dgrad / pn2_conv1x1_wgrad with identity BN coefficients."""

    @staticmethod
    def forward(ctx, rows, weight, bias, padded=False):
        lib, st = _lib.load(), _lib.stream()
        P, ldx = rows.shape
        co = weight.shape[0]
        ci = weight.numel() // co
        y = _empty_rows(P, co, rows.device)
        _check(lib.pn2_conv1x1_fwd(_p(rows), ldx, None, _p(_contig_weight(weight)), ci, _p(bias), _p(y), y.shape[1], P, ci, co,
                                   None, None, None, st), "pn2_conv1x1_fwd")
        ctx.save_for_backward(rows, weight)
        ctx.params = (weight, bias)             # leaf parameters (no grad_fn): no reference cycle
        ctx.dims = (P, ci, co, ldx, y.shape[1])
        ctx.padded = bool(padded)
        # padded: the [P, round4(co)] matrix as the GEMM wrote it (pad columns zero), for a consumer that reads the padded
        # layout itself (log_softmax_rows) -- the gradient then comes back padded too, with no zero fill + slice copy
        return y if (padded or y.shape[1] == co) else y[:, :co]

    @staticmethod
    def backward(ctx, grad):
        lib, st = _lib.load(), _lib.stream()
        rows, weight = ctx.saved_tensors
        P, ci, co, ldx, ldy = ctx.dims
        dev = rows.device
        if ldy != co and not ctx.padded:
            g = torch.zeros(P, ldy, device=dev, dtype=torch.float32)
            g[:, :co] = grad
        else:
            g = grad.contiguous()
        ident = _ident_coef(co, dev)                       # dY := 1*g + 0*(y - 0) + 0
        direct = _direct_ok(ctx.params)                    # the kernel's atomics add straight into .grad (see set_direct_grad_accumulation)
        if direct:
            dW, db = ctx.params[0].grad.view(co, ci), ctx.params[1].grad
        else:
            zb = _zeros_small(4 * (co * ci + co), dev).view(torch.float32)
            dW = zb[:co * ci].view(co, ci)
            db = zb[co * ci:]
        _check(lib.pn2_conv1x1_wgrad(_p(g), ldy, None, 0, None, 0, _p(g), ldy, _p(ident), _p(rows), ldx, None, _p(dW), ci,
                                     _p(db), P, co, ci, None, st), "pn2_conv1x1_wgrad")
        d_rows = None
        if ctx.needs_input_grad[0]:
            d_rows = torch.empty(P, ldx, device=dev, dtype=torch.float32)     # pad lanes written (0) by the GEMM
            _check(lib.pn2_conv1x1_dgrad(_p(g), ldy, None, 0, None, 0, _p(g), ldy, _p(ident), _p(_contig_weight(weight)), ci,
                                         None, 0, None,
                                         _p(d_rows), ldx, None, P, co, ci, None, None, st), "pn2_conv1x1_dgrad")
        if direct:
            return d_rows, None, None, None
        return d_rows, dW.view_as(weight), db, None


def conv1x1(rows, conv, padded=False):
    """rows [P, round4(C_in)] -> [P, C_out] through ``conv`` (an nn.Conv1d/Conv2d with kernel size 1), HIP kernels.
    ``padded``: return the [P, round4(C_out)] matrix as the GEMM wrote it (for ``log_softmax_rows``)."""
    rows = _gpu_f32(rows, "rows")
    return _Conv1x1.apply(rows, conv.weight, conv.bias, padded)


class _LogSoftmaxRows(torch.autograd.Function):
    """F.log_softmax(x[:, :C], dim=-1) (model/pointnet2.py:175) on the padded logits [P, round4(C)] of ``conv1x1(...,
    padded=True)``: one HIP launch each way, the gradient is handed back in the padded layout."""

    @staticmethod
    def forward(ctx, x, C):
        out = torch.empty(x.shape[0], C, device=x.device, dtype=torch.float32)
        _check(_lib.load().pn2_log_softmax_fwd(_p(x), x.shape[1], x.shape[0], C, _p(out), C, _lib.stream()), "pn2_log_softmax_fwd")
        ctx.save_for_backward(out)
        ctx.ld = x.shape[1]
        return out

    @staticmethod
    def backward(ctx, grad):
        out, = ctx.saved_tensors
        P, C = out.shape
        if grad.stride(-1) != 1:
            grad = grad.contiguous()
        gx = torch.empty(P, ctx.ld, device=out.device, dtype=torch.float32)
        _check(_lib.load().pn2_log_softmax_bwd(_p(grad), grad.stride(0), _p(out), C, P, C, _p(gx), ctx.ld, _lib.stream()),
               "pn2_log_softmax_bwd")
        return gx, None


def log_softmax_rows(x_padded, C):
    """[P, round4(C)] padded logits -> [P, C] log-probabilities (HIP); falls back to ATen beyond 64 classes."""
    if C > 64 or x_padded.shape[1] > 64:
        return torch.nn.functional.log_softmax(x_padded[:, :C], dim=-1)
    return _LogSoftmaxRows.apply(_gpu_f32(x_padded, "logits"), C)


def _flat_params(convs, bns):
    flat, cfg = [], []
    for conv, bn in zip(convs, bns):
        if bn.momentum is None:
            raise NotImplementedError("cumulative-average BatchNorm (momentum=None) is not supported")
        flat += [conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked]
        cfg.append((float(bn.eps), float(bn.momentum)))
    return flat, tuple(cfg)


def shared_mlp(rows, c_in, convs, bns, pool, training, dest=None):
    """rows [P, ld] -> [P/pool, C_out] (pool > 0) or [P, C_out] (pool == 0) through the HIP kernels.
    ``dest`` (training, pooled): see _SharedMLP."""
    flat, cfg = _flat_params(convs, bns)
    rows = _gpu_f32(rows, "rows")
    if rows.dim() != 2 or rows.shape[1] != _r4(c_in):
        raise RuntimeError("rows must be [P, round4(c_in)] with zero pad columns")
    if not training and _fused_eval_ok(c_in, convs, pool) and (pool == 0 or rows.shape[0] % pool == 0):
        return fused_eval_rows(rows, c_in, convs, bns, pool)
    return _SharedMLP.apply(rows, c_in, pool, training, cfg, None, dest, *flat)


# ------------------------------------------------------------------------------------- eval-mode fused module
# Under .eval() BatchNorm is a fixed affine map: it folds into the conv in front of it and a whole module becomes ONE
# launch (csrc/eval.hip: gather -> L x (GEMM + ReLU) -> max, activations in LDS).  Taken when no gradient is wanted
# (torch.no_grad() / inference_mode: the reference's viewer loop, pcdvis.py:118-136); PN2_FUSED_EVAL=0 keeps the
# layer-by-layer kernels (A/B runs, and the path eval-mode autograd uses).
FUSED_EVAL = os.environ.get("PN2_FUSED_EVAL", "1") == "1"
_fold_cache = weakref.WeakKeyDictionary()       # first conv of a stack -> fold entry (dies with the module: no id() aliasing)
_PARAM_GEN = [0]


def bump_param_generation():
    """Tell the eval-mode fold cache that parameters or BatchNorm buffers MAY have been written behind autograd's back.

    ``tensor._version`` only counts writes made through torch: the HIP library updates running statistics
    (pn2_bn_finalize), parameters (pn2_adam_step) and everything a replayed hipGraph touches through raw pointers.
    Every such path calls this (``_SharedMLP.forward`` in training mode, ``optim.Adam.step``, ``GraphedStep.replay``,
    ``parallel.broadcast_module``); a fold made under an older generation is redone on its next use."""
    _PARAM_GEN[0] += 1


_DATA_GEN = [0]


def bump_data_generation():
    """Tell the channel-last memo (``_channel_last``) that INPUT tensors may have been rewritten behind autograd's back.

    ``tensor._version`` only counts writes made through torch; ``loader.prepare_batch(out=...)`` (pn2_prepare_clouds into
    static buffers) and a replayed hipGraph refill tensors through raw pointers.  A channel-first view of such a buffer that a
    caller holds across steps would otherwise get the PREVIOUS batch's copy back on the next eager forward."""
    _DATA_GEN[0] += 1


def _folded_layers(convs, bns):
    """(ctypes array of pn2_eval_layer, tensors kept alive) with W' = diag(gamma / sqrt(var + eps)) W and
    b' = (b - mean) * gamma / sqrt(var + eps) + beta, computed in fp64, rows padded to round8(C_in).

    Cached per stack.  An entry is fresh while the parameter generation (see bump_param_generation) and the autograd
    versions of all six tensors per layer are unchanged; a stale entry is refreshed IN PLACE, so a hipGraph that captured a
    pn2_fused_eval launch keeps reading valid memory and sees the new weights on its next replay (the pointers baked into
    the graph never change for the lifetime of the module)."""
    ids = tuple(id(m) for m in convs) + tuple(id(m) for m in bns)
    ver = tuple(int(t._version) for conv, bn in zip(convs, bns)
                for t in (conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var))
    shapes = tuple((conv.weight.shape[0], conv.weight.numel() // conv.weight.shape[0], conv.weight.device) for conv in convs)
    hit = _fold_cache.get(convs[0])
    if hit is not None and hit["ids"] == ids and hit["shapes"] == shapes:
        if hit["ver"] == ver and hit["gen"] == _PARAM_GEN[0]:
            return hit["arr"], hit["keep"]
        keep, arr = hit["keep"], hit["arr"]
    else:
        hit, keep, arr = None, [], (_lib.EvalLayer * len(convs))()
    for l, (conv, bn) in enumerate(zip(convs, bns)):
        co = conv.weight.shape[0]
        ci = conv.weight.numel() // co
        scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
        w = conv.weight.detach().double().reshape(co, ci) * scale[:, None]
        b = (conv.bias.detach().double() - bn.running_mean.detach().double()) * scale + bn.bias.detach().double()
        ldw = (ci + 7) & ~7
        if hit is None:
            wp = torch.zeros(co, ldw, device=w.device, dtype=torch.float32)
            bp = torch.empty(co, device=w.device, dtype=torch.float32)
            keep += [wp, bp]
            arr[l].W, arr[l].bias, arr[l].K, arr[l].N, arr[l].ldw = wp.data_ptr(), bp.data_ptr(), ci, co, ldw
        else:
            wp, bp = keep[2 * l], keep[2 * l + 1]
        wp[:, :ci].copy_(w)                        # pad columns stay zero
        bp.copy_(b)
    _fold_cache[convs[0]] = {"ids": ids, "shapes": shapes, "ver": ver, "gen": _PARAM_GEN[0], "arr": arr, "keep": keep}
    return arr, keep


def _fused_eval_ok(c_in, convs, pool):
    if not FUSED_EVAL or torch.is_grad_enabled() or len(convs) > 4:
        return False
    if pool not in (0, 16) and pool % 32:
        return False
    widths = [c_in] + [c.weight.shape[0] for c in convs[:-1]]
    return max((w + 7) & ~7 for w in widths) + 4 <= 160 * 1024 // (2 * 32 * 4)


def fused_eval_rows(rows, c_in, convs, bns, pool):
    """Plain rows [P, ld] -> [P / pool, C_L] (pool > 0) or [P, C_L] through pn2_fused_eval."""
    arr, _keep = _folded_layers(convs, bns)
    P = rows.shape[0]
    cl = convs[-1].weight.shape[0]
    out = torch.empty(P // pool if pool else P, _r4(cl), device=rows.device, dtype=torch.float32)
    _check(_lib.load().pn2_fused_eval(_p(rows), rows.shape[1], None, None, None, None, P, 0, 0, pool or 1, 0, 1,
                                      ctypes.cast(arr, ctypes.c_void_p), len(convs), pool, _p(out), out.shape[1], _lib.stream()),
           "pn2_fused_eval")
    return out[:, :cl] if out.shape[1] != cl else out


def fused_eval_grouped(xyz, points, new_xyz, idx, S, K, xyz_first, convs, bns):
    """Gather + centre + concat + MLP + max over the K neighbours -> [B*S, C_L]; idx None: group_all."""
    arr, _keep = _folded_layers(convs, bns)
    B, N, _ = xyz.shape
    D = 0 if points is None else points.shape[2]
    cl = convs[-1].weight.shape[0]
    out = torch.empty(B * S, _r4(cl), device=xyz.device, dtype=torch.float32)
    _check(_lib.load().pn2_fused_eval(None, 0, _p(xyz), _p(points), _p(new_xyz), _p(idx), B, N, S, K, D, int(xyz_first),
                                      ctypes.cast(arr, ctypes.c_void_p), len(convs), K, _p(out), out.shape[1], _lib.stream()),
           "pn2_fused_eval")
    return out[:, :cl] if out.shape[1] != cl else out


FACTORISE_MIN_FEATURES = 32      # below this the grouped rows are narrower than the gathered layer-1 output


def _factorised(D, n_layers, training):
    return D >= FACTORISE_MIN_FEATURES and n_layers >= 2 and training


def _group_inverse(idx, N, D, n_layers, training):
    """Inverse of a ball-query index [B,S,K] over the N source points, when the factorised first layer's backward
    will want it (same predicate in record and replay mode: the tape order must not depend on the mode)."""
    if not (GATHER_BACKWARD and _factorised(D, n_layers, training)):
        return None
    B, S, K = idx.shape
    return _inverse_index(idx.view(B, S * K), N)


def grouped_mlp(xyz, points, new_xyz, idx, xyz_first, convs, bns, training, inv=None, dest=None):
    """Group + shared MLP + max over the K neighbours of one SA scale -> [B*S, C_out].

    With enough input features (and a trainable stack of >= 2 layers) the first layer runs factorised over
    the source points (see _SharedMLP); otherwise the grouped [P, 3+D] matrix is built and fed to the GEMMs.
    """
    B, S, K = idx.shape
    D = 0 if points is None else points.shape[2]
    if not training and _fused_eval_ok(3 + D, convs, K):
        return fused_eval_grouped(xyz, points, new_xyz, idx.contiguous(), S, K, xyz_first, convs, bns)
    if _factorised(D, len(convs), training):
        flat, cfg = _flat_params(convs, bns)
        return _SharedMLP.apply(points, 3 + D, K, training, cfg, (xyz, new_xyz, idx, xyz_first, inv), dest, *flat)
    if (GATHER_CONV and not FUSED_BN_TAILS and 1 <= D <= 9 and convs[0].out_channels in (32, 64) and (B * S * K) % 64 == 0 and new_xyz is not None and
            idx is not None):
        # narrow first layer (the sa1 stacks): pn2_group and the first conv are one launch (pn2_group_conv_fwd)
        flat, cfg = _flat_params(convs, bns)
        return _SharedMLP.apply(points, 3 + D, K, training, cfg, (xyz, new_xyz, idx.contiguous(), xyz_first, None, True), dest, *flat)
    rows = _Group.apply(xyz, points, new_xyz, idx, S, K, xyz_first)
    return shared_mlp(rows, 3 + D, convs, bns, K, training, dest)


# --------------------------------------------------------------------------------------- grouping API

def _channel_last(t, name):
    """[B,C,N] (any strides) -> contiguous [B,N,C]; free when t is a channel-first view of channel-last storage.

    A caller tensor that really is channel-first (the network input: xyz and features, consumed by sa1 AND by fp1) is copied
    once: the copy rides on the tensor object (``_pn2_rows``, with the version counter AND the data generation it was made
    from -- raw-pointer writers bump the latter, see ``bump_data_generation``) and the second consumer gets the same tensor
    back -- autograd then accumulates both gradients into it as for any shared value."""
    v = t.permute(0, 2, 1)
    capturing = torch.cuda.is_current_stream_capturing()
    # (a differentiable copy belongs to ONE autograd graph: never kept; a capture that did not announce its scope must bake in
    # nothing that an eager call made or could later replace)
    if v.is_contiguous() or t.requires_grad or (capturing and _capture_scope is None):
        return _gpu_f32(v, name)
    memo = getattr(t, "_pn2_rows", None)
    key = (t._version, _DATA_GEN[0], t.data_ptr(), tuple(t.stride()))
    if memo is not None and memo[0] == key and memo[1] is (_capture_scope if capturing else None) and \
            memo[2] == torch.is_grad_enabled():
        return memo[3]
    r = _gpu_f32(v, name)
    try:
        t._pn2_rows = (key, _capture_scope if capturing else None, torch.is_grad_enabled(), r)
    except Exception:                     # (objects that refuse attributes: plain copy every time)
        pass
    return r


def sample_and_group(npoint, radius, nsample, xyz, points, returnfps=False):
    """pointnet_util.py:110-137: -> new_xyz [B,S,3], new_points [B,S,K,3+D] (xyz first)."""
    xyz = _gpu_f32(xyz, "xyz")
    B, N, C = xyz.shape
    fps_idx = farthest_point_sample(xyz, npoint)
    new_xyz = index_points(xyz, fps_idx, _checked=False)
    idx = query_ball_point(radius, nsample, xyz, new_xyz)
    grouped_xyz = index_points(xyz, idx)
    pts = None if points is None else _gpu_f32(points, "points")
    rows = _Group.apply(xyz, pts, new_xyz, idx, npoint, nsample, True)
    D = 0 if pts is None else pts.shape[2]
    new_points = rows.view(B, npoint, nsample, -1)[..., :3 + D]
    if returnfps:
        return new_xyz, new_points, grouped_xyz, fps_idx
    return new_xyz, new_points


def sample_and_group_all(xyz, points):
    """pointnet_util.py:140-157: -> zeros [B,1,3], [B,1,N,3+D] (xyz NOT centred)."""
    xyz = _gpu_f32(xyz, "xyz")
    B, N, C = xyz.shape
    new_xyz = torch.zeros(B, 1, C, device=xyz.device)          # a fresh tensor: the caller of the public function owns it
    pts = None if points is None else _gpu_f32(points, "points")
    rows = _Group.apply(xyz, pts, None, None, 1, N, True)
    D = 0 if pts is None else pts.shape[2]
    return new_xyz, rows.view(B, 1, N, -1)[..., :3 + D]


# --------------------------------------------------------------------------------------- modules

class PointNetSetAbstraction(nn.Module):
    """Drop-in for model/pointnet_util.py:160-201 (same constructor, forward and state_dict)."""

    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all):
        super().__init__()
        self.npoint = npoint
        self.radius = radius
        self.nsample = nsample
        self.mlp_convs = nn.ModuleList()      # parameter containers only: their forward is never called
        self.mlp_bns = nn.ModuleList()
        last_channel = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv2d(last_channel, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm2d(out_channel))
            last_channel = out_channel
        self.group_all = group_all
        self.in_channel = in_channel

    def forward(self, xyz, points, fps_start=None):
        """xyz [B,3,N], points [B,D,N] or None -> new_xyz [B,3,S], new_points [B,C',S]."""
        xyz = _channel_last(xyz, "xyz")
        pts = None if points is None else _channel_last(points, "points")
        B, N, _ = xyz.shape
        if self.group_all:
            new_xyz = _zero_centres(B, 3, xyz.device)
            if not _recording() and not self.training and _fused_eval_ok(3 + (0 if pts is None else pts.shape[2]), self.mlp_convs, N):
                out = fused_eval_grouped(xyz, pts, None, None, 1, N, True, self.mlp_convs, self.mlp_bns)
                return new_xyz.permute(0, 2, 1), out.view(B, 1, -1).permute(0, 2, 1)
            rows = None if _recording() else _Group.apply(xyz, pts, None, None, 1, N, True)
            S, K = 1, N
        else:
            S, K = self.npoint, self.nsample
            fps_idx = farthest_point_sample(xyz, S, fps_start)
            new_xyz = _sampled_centres(xyz, fps_idx)
            idx = query_ball_point(self.radius, K, xyz, new_xyz)
            inv = _group_inverse(idx, N, 0 if pts is None else pts.shape[2], len(self.mlp_convs), self.training)
            rows = None
        c_in = 3 + (0 if pts is None else pts.shape[2])
        if c_in != self.in_channel:
            raise RuntimeError("expected %d input channels (3 + features), got %d" % (self.in_channel, c_in))
        if _recording():
            return new_xyz.permute(0, 2, 1), _placeholder(B, self.mlp_convs[-1].out_channels, S, xyz.device)
        if rows is None:
            out = grouped_mlp(xyz, pts, new_xyz, idx, True, self.mlp_convs, self.mlp_bns, self.training, inv)
        else:
            out = shared_mlp(rows, c_in, self.mlp_convs, self.mlp_bns, K, self.training)
        return new_xyz.permute(0, 2, 1), out.view(B, S, -1).permute(0, 2, 1)


class PointNetSetAbstractionMsg(nn.Module):
    """Drop-in for model/pointnet_util.py:204-261."""

    def __init__(self, npoint, radius_list, nsample_list, in_channel, mlp_list):
        super().__init__()
        self.npoint = npoint
        self.radius_list = radius_list
        self.nsample_list = nsample_list
        self.conv_blocks = nn.ModuleList()
        self.bn_blocks = nn.ModuleList()
        for i in range(len(mlp_list)):
            convs = nn.ModuleList()
            bns = nn.ModuleList()
            last_channel = in_channel + 3
            for out_channel in mlp_list[i]:
                convs.append(nn.Conv2d(last_channel, out_channel, 1))
                bns.append(nn.BatchNorm2d(out_channel))
                last_channel = out_channel
            self.conv_blocks.append(convs)
            self.bn_blocks.append(bns)
        self.in_channel = in_channel

    def forward(self, xyz, points, fps_start=None):
        xyz = _channel_last(xyz, "xyz")
        pts = None if points is None else _channel_last(points, "points")
        B, N, _ = xyz.shape
        S = self.npoint
        new_xyz = _sampled_centres(xyz, farthest_point_sample(xyz, S, fps_start))
        c_in = 3 + (0 if pts is None else pts.shape[2])
        outs = []
        # The scales are independent until the final concatenation: with MSG_SCALE_STREAMS on, each one is issued
        # on its own HIP stream (autograd replays the backward on the same streams), so under graph capture they
        # become parallel branches and one scale's small kernels / GEMM tails fill under another's GEMMs.
        branch = MSG_SCALE_STREAMS and not _recording() and len(self.radius_list) > 1
        if branch:
            main = torch.cuda.current_stream(xyz.device)
            streams = _scale_streams(xyz.device, len(self.radius_list))
        n_scales = len(self.radius_list)
        # Every scale writes its pooled output straight into its column slice of ONE [B*S, sum C] matrix (the layout
        # torch.cat(new_points_list, dim=1) of pointnet_util.py:260 produces): no concatenation copy in the forward, and the
        # backward hands each scale its slice of the incoming gradient in place.
        widths = [c[-1].out_channels for c in self.conv_blocks]
        big = None
        if not _recording() and self.training and MSG_CONCAT_IN_PLACE and all(w % 4 == 0 for w in widths):
            big = torch.empty(B * S, sum(widths), device=xyz.device, dtype=torch.float32)
        col = 0
        for i, radius in enumerate(self.radius_list):
            K = self.nsample_list[i]
            # the last scale stays on the calling stream: one branch fewer (a captured step then has four concurrent
            # branches with the geometry prefetch -- as many as the graph executor has hardware queues)
            on_side = branch and not (MSG_LAST_SCALE_ON_MAIN and i == n_scales - 1)
            if on_side:
                streams[i].wait_stream(main)
                for t in (xyz, pts, new_xyz, big):         # allocated on the main stream, used on the branch
                    if t is not None:
                        t.record_stream(streams[i])
                ctx_mgr = torch.cuda.stream(streams[i])
            else:
                ctx_mgr = contextlib.nullcontext()
            with ctx_mgr:
                idx = query_ball_point(radius, K, xyz, new_xyz)
                inv = _group_inverse(idx, N, 0 if pts is None else pts.shape[2], len(self.conv_blocks[i]), self.training)
                if _recording():
                    continue
                outs.append(grouped_mlp(xyz, pts, new_xyz, idx, False, self.conv_blocks[i], self.bn_blocks[i],
                                        self.training, inv, None if big is None else (big, col)))   # features first (:247)
            col += widths[i]
        if branch:
            for i, (st, o) in enumerate(zip(streams, outs)):
                if MSG_LAST_SCALE_ON_MAIN and i == n_scales - 1:
                    continue
                main.wait_stream(st)
                o.record_stream(main)                      # produced on the branch, concatenated on the main stream
        if _recording():
            return new_xyz.permute(0, 2, 1), _placeholder(B, sum(c[-1].out_channels for c in self.conv_blocks), S, xyz.device)
        in_place = big is not None and all(o.data_ptr() == big.data_ptr() + 4 * c0 and o.stride(0) == big.shape[1]
                                           for o, c0 in zip(outs, np.cumsum([0] + widths[:-1])))
        out = _ConcatView.apply(big, *outs) if in_place else torch.cat(outs, dim=1)       # [B*S, sum C]
        return new_xyz.permute(0, 2, 1), out.view(B, S, -1).permute(0, 2, 1)


class _ConcatView(torch.autograd.Function):
    """The concatenation of tensors that already ARE the column slices of ``big``: forward returns ``big`` (no copy), backward
    hands every input its column slice of the gradient (views: the pooled backward reads them at their pitch)."""

    @staticmethod
    def forward(ctx, big, *parts):
        ctx.widths = [p.shape[1] for p in parts]
        return big.view(big.shape)

    @staticmethod
    def backward(ctx, grad):
        outs, c0 = [], 0
        for w in ctx.widths:
            outs.append(grad[:, c0:c0 + w])
            c0 += w
        return (None,) + tuple(outs)


class PointNetFeaturePropagation(nn.Module):
    """Drop-in for model/pointnet_util.py:264-313."""

    def __init__(self, in_channel, mlp):
        super().__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last_channel = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv1d(last_channel, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm1d(out_channel))
            last_channel = out_channel
        self.in_channel = in_channel

    def forward(self, xyz1, xyz2, points1, points2):
        """xyz1 [B,3,N], xyz2 [B,3,S], points1 [B,D1,N] or None, points2 [B,D2,S] -> [B,D',N]."""
        x1 = _channel_last(xyz1, "xyz1")
        x2 = _channel_last(xyz2, "xyz2")
        p2 = _channel_last(points2, "points2")
        p1 = None if points1 is None else _channel_last(points1, "points1")
        B, N, _ = x1.shape
        S = x2.shape[1]
        want_inv = GATHER_BACKWARD and self.training and S != 1
        if _recording():
            if S != 1:
                idx, _, _ = three_nn(x1, x2)
                if want_inv:
                    _inverse_index(idx.view(B, N * 3), S)
            return _placeholder(B, self.mlp_convs[-1].out_channels, N, x1.device)
        if S == 1:                                      # pointnet_util.py:292-293: broadcast the single feature row
            # = interpolation from "neighbours" (0, 0, 0) with weights (1, 0, 0): 1 * p + 0 * p + 0 * p is p exactly, so the
            # same launch that copies points1 and interpolates for S > 1 writes the concatenated rows here too (was: expand +
            # cat + pad + contiguous in ATen, and a sum in the backward)
            idx, w, _, _ = _broadcast_neighbours(B, N, x1.device, False)
            rows = _InterpCat.apply(p1, p2, idx, w, None, None)      # (backward: pn2_three_interp_bwd's S == 1 column sum)
            c_in = p2.shape[2] + (0 if p1 is None else p1.shape[2])
        else:
            idx, _, w = three_nn(x1, x2)
            inv = _inverse_index(idx.view(B, N * 3), S) if want_inv else (None, None)
            rows = _InterpCat.apply(p1, p2, idx, w, inv[0], inv[1])
            c_in = p2.shape[2] + (0 if p1 is None else p1.shape[2])
        if c_in != self.in_channel:
            raise RuntimeError("expected %d input channels, got %d" % (self.in_channel, c_in))
        out = shared_mlp(rows, c_in, self.mlp_convs, self.mlp_bns, 0, self.training)
        return out.view(B, N, -1).permute(0, 2, 1)
