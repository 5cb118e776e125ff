"""ctypes binding of libpn2_hip.so (the C ABI declared in include/pn2.h).

The HIP library is the only compute path of this package: if it is missing or a symbol
cannot be resolved this module raises -- there is no CPU or eager-PyTorch fallback.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PN2_LIB_PATH") or os.path.join(_HERE, "libpn2_hip.so")   # override: A/B runs of two builds
CSRC = os.path.join(_HERE, "csrc")

_vp, _i, _i64, _f, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double

# name -> (restype, argtypes); mirrors include/pn2.h one to one (tests check the header against this table)
SIGNATURES = {
    "pn2_version": (_i, []),
    "pn2_error_string": (ctypes.c_char_p, [_i]),
    "pn2_last_kernel": (ctypes.c_char_p, []),
    "pn2_clear_last_kernel": (None, []),
    "pn2_set_option": (_i, [ctypes.c_char_p, _i]),
    "pn2_get_option": (_i, [ctypes.c_char_p, _vp]),
    "pn2_option_name": (ctypes.c_char_p, [_i]),
    "pn2_fps_workspace_bytes": (_i64, [_i, _i, _i]),
    "pn2_fps": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "pn2_ball_query": (_i, [_vp, _vp, _i, _i, _i, _f, _i, _vp, _vp]),
    "pn2_ball_query_workspace_bytes": (_i64, [_i, _i, _i]),
    "pn2_ball_query_ws": (_i, [_vp, _vp, _i, _i, _i, _f, _i, _vp, _vp, _vp]),
    "pn2_square_distance": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "pn2_three_nn": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "pn2_gather_rows": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "pn2_gather_rows_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "pn2_group": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "pn2_group_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "pn2_group_affine_fwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "pn2_group_affine_bwd": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    "pn2_three_interp": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp]),
    "pn2_three_interp_bwd": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "pn2_copy_cols": (_i, [_vp, _i, _i, _vp, _i, _i, _i64, _i, _vp]),
    "pn2_conv1x1_fwd": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i64, _i, _i, _vp, _vp, _vp, _vp]),
    "pn2_group_conv_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _vp, _vp]),
    "pn2_conv1x1_fwd_pool": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i64, _i, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "pn2_bn_pool_select": (_i, [_vp, _vp, _i64, _i, _vp, _i, _vp, _vp, _vp]),
    "pn2_bn_finalize": (_i, [_vp, _i64, _i, _vp, _vp, _f, _f, _i, _vp, _vp, _vp, _vp, _vp]),
    "pn2_bn_relu_max": (_i, [_vp, _i, _vp, _i64, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "pn2_pool_bwd_reduce": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp]),
    "pn2_pool_bwd_reduce_ld": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp]),
    "pn2_pool_bwd_reduce_rec": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _vp]),
    "pn2_relu_bwd_reduce": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i64, _i, _vp, _i, _vp, _vp, _vp]),
    "pn2_bn_bwd_coef": (_i, [_vp, _i64, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp]),
    "pn2_conv1x1_dgrad": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _vp,
                               _i64, _i, _i, _vp, _vp, _vp]),
    "pn2_conv1x1_wgrad": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp,
                               _i64, _i, _i, _vp, _vp]),
    "pn2_conv1x1_wgrad_workspace_bytes": (_i64, [_i64, _i, _i, _i]),
    "pn2_conv1x1_wgrad_ws": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp,
                                  _i64, _i, _i, _vp, _vp, _vp]),
    "pn2_conv1x1_wgrad_cf_scratch_bytes": (_i64, []),
    "pn2_conv1x1_wgrad_cf": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i64, _i, _i, _vp, _vp]),
    "pn2_conv1x1_bwd_pair": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i,
                                  _i64, _i, _i, _vp, _vp]),
    "pn2_res_supported": (_i, [_i64, _i, _i]),
    "pn2_bwd_res_supported": (_i, [_i64, _i, _i, _i, _i]),
    "pn2_conv1x1_bwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i,
                             _i64, _i, _i, _vp, _vp]),
    "pn2_conv1x1_bwd_cf_supported": (_i, [_i64, _i, _i, _i]),
    "pn2_conv1x1_bwd_cf_scratch_bytes": (_i64, [_i, _i]),
    "pn2_conv1x1_bwd_cf": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i64, _i, _i, _vp, _vp, _vp]),
    "pn2_conv1x1_bwd_first_supported": (_i, [_i64, _i, _i, _i]),
    "pn2_conv1x1_bwd_first": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _vp, _i64, _i, _i, _vp, _vp]),
    "pn2_fused_eval": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _i, _vp]),
    "pn2_invert_index": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "pn2_three_interp_bwd_seg": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "pn2_group_affine_bwd_seg": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _i,
                                         _vp, _vp, _vp]),
    "pn2_nll_loss_workspace_bytes": (_i64, [_i64]),
    "pn2_nll_loss_fwd": (_i, [_vp, _i, _vp, _vp, _i64, _i, _i64, _vp, _vp, _vp, _vp]),
    "pn2_nll_loss_bwd": (_i, [_vp, _vp, _i64, _i, _i64, _vp, _vp, _vp, _i, _vp]),
    "pn2_log_softmax_fwd": (_i, [_vp, _i, _i64, _i, _vp, _i, _vp]),
    "pn2_log_softmax_bwd": (_i, [_vp, _i, _vp, _i, _i64, _i, _vp, _i, _vp]),
    "pn2_adam_step": (_i, [_vp, _vp, _vp, _vp, _i64, _d, _d, _d, _d, _d, _i64, _vp, _vp, _i, _vp]),
    "pn2_prepare_clouds": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
}


ABI_VERSION = 11
PN2_EUNSUPPORTED = -3            # include/pn2.h
PN2_OK_SPLIT = 1                 # pn2_conv1x1_bwd_pair: done as two launches
DWX_REPLICAS = 32        # PN2_DWX_REPLICAS of include/pn2.h


class BnFinalizeTail(ctypes.Structure):
    """pn2_bn_finalize_tail of include/pn2.h."""
    _fields_ = [("ticket", _vp), ("gamma", _vp), ("beta", _vp), ("eps", _f), ("momentum", _f), ("running_mean", _vp),
                ("running_var", _vp), ("num_batches_tracked", _vp), ("affine", _vp)]


class BnCoefTail(ctypes.Structure):
    """pn2_bn_coef_tail of include/pn2.h."""
    _fields_ = [("ticket", _vp), ("gamma", _vp), ("affine", _vp), ("use_batch_stats", _i), ("coef", _vp), ("dgamma", _vp),
                ("dbeta", _vp), ("accumulate", _i)]


class BnLazy(ctypes.Structure):
    """pn2_bn_lazy of include/pn2.h: statistics -> affine block inside the first launch that reads the block."""
    _fields_ = [("stats", _vp), ("gamma", _vp), ("beta", _vp), ("eps", _f), ("momentum", _f), ("running_mean", _vp),
                ("running_var", _vp), ("num_batches_tracked", _vp), ("affine", _vp), ("count", _i64), ("C", _i)]


class BnCoefLazy(ctypes.Structure):
    """pn2_bn_coef_lazy of include/pn2.h: reductions -> backward coefficients inside the first launch that reads them."""
    _fields_ = [("red", _vp), ("gamma", _vp), ("affine", _vp), ("coef", _vp), ("dgamma", _vp), ("dbeta", _vp),
                ("accumulate", _i), ("count", _i64), ("C", _i)]


class EvalLayer(ctypes.Structure):
    """pn2_eval_layer of include/pn2.h."""
    _fields_ = [("W", _vp), ("bias", _vp), ("K", _i), ("N", _i), ("ldw", _i)]


class Pn2Error(RuntimeError):
    pass


def build(force=False):
    """Compile libpn2_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "-s", "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j4"])
    return LIB_PATH


_lib = None
_raw = None
_profile = None          # None, or a list receiving (name, args, start_event, end_event) per call


class _Timed:
    """Proxy handed out while a call profile is active: brackets every launch with HIP events recorded on
    the stream the kernel is launched on (the handle passed as the last argument of every launch entry point)."""

    def __getattr__(self, name):
        fn = getattr(_raw, name)
        if not name.startswith("pn2_") or name in ("pn2_version", "pn2_error_string", "pn2_set_option", "pn2_get_option", "pn2_option_name", "pn2_fps_workspace_bytes",
                                                   "pn2_nll_loss_workspace_bytes", "pn2_res_supported", "pn2_bwd_res_supported", "pn2_conv1x1_wgrad_workspace_bytes",
                                                   "pn2_conv1x1_wgrad_cf_scratch_bytes", "pn2_conv1x1_bwd_cf_supported", "pn2_conv1x1_bwd_first_supported", "pn2_conv1x1_bwd_cf_scratch_bytes", "pn2_last_kernel", "pn2_clear_last_kernel",
                                                   "pn2_ball_query_workspace_bytes"):
            return fn

        def timed(*args):
            import torch
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            h = args[-1]
            s = torch.cuda.ExternalStream(h) if h else torch.cuda.default_stream()
            _raw.pn2_clear_last_kernel()
            a.record(s)
            rc = fn(*args)
            b.record(s)
            if rc != PN2_EUNSUPPORTED:         # (a refused shape launched nothing: the caller takes its other route)
                # (pn2_conv1x1_bwd_pair that ran as two launches is booked under a name of its own: bench.py splits it)
                kern = _raw.pn2_last_kernel()      # the GEMM launchers say which template instantiation they enqueued (or None)
                _profile.append((name + "_split" if (name == "pn2_conv1x1_bwd_pair" and rc == PN2_OK_SPLIT) else name, args, a, b,
                                 kern.decode() if kern else None))
            return rc
        return timed


class call_profile:
    """``with call_profile() as calls:`` -> list of (entry point, args, start, end, kernel name or None) for every C-ABI call made
    inside; ``start.elapsed_time(end)`` after a synchronize gives that launch's device time in ms."""

    def __enter__(self):
        global _lib, _profile
        load()
        _profile = []
        _lib = _Timed()
        return _profile

    def __exit__(self, *exc):
        global _lib, _profile
        _lib = _raw
        _profile = None
        return False


def load():
    """Load the library and bind every symbol of SIGNATURES; raises if anything is missing."""
    global _lib, _raw
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Pn2Error("libpn2_hip.so not found at %s -- build it with `python -c \"import __graft_entry__ as g; "
                       "g.build()\"` (hipcc, gfx950). There is no fallback path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.pn2_version() != ABI_VERSION:
        raise Pn2Error("libpn2_hip.so ABI version %d, expected %d -- rebuild it" % (lib.pn2_version(), ABI_VERSION))
    forward_env_options(lib)         # (before the library is published: a malformed PN2_<OPTION> raises on EVERY load, not only the first)
    _lib = _raw = lib
    return lib


def options(lib=None):
    """{name: value} of every library option (pn2_option_name / pn2_get_option)."""
    lib = lib or load()
    out, i = {}, 0
    while True:
        name = lib.pn2_option_name(i)
        if name is None:
            return out
        v = ctypes.c_int(0)
        if lib.pn2_get_option(name, ctypes.addressof(v)) == 0:
            out[name.decode()] = v.value
        i += 1


def set_option(name, value):
    """pn2_set_option; raises on an unknown name."""
    check(load().pn2_set_option(name.encode(), int(value)), "pn2_set_option(%s)" % name)


def forward_env_options(lib):
    """The library itself never reads the environment (include/pn2.h, options): A/B runs set PN2_<OPTION>=<int> in the
    environment of THIS binding, which hands them to pn2_set_option once at load time."""
    for name in options(lib):
        v = os.environ.get(name)
        if v is not None:
            try:
                value = int(v)
            except ValueError:
                raise Pn2Error("%s=%r is not an integer" % (name, v))
            rc = lib.pn2_set_option(name.encode(), value)
            if rc != 0:
                raise Pn2Error("pn2_set_option(%s, %d) failed: %d" % (name, value, rc))


def check(rc, what):
    if rc != 0:
        raise Pn2Error("%s failed: %s (%d)" % (what, load().pn2_error_string(rc).decode(), rc))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
