"""Data parallelism for the PointNet++ path: one process per GPU, one flat gradient all-reduce per step.

The reference scales with ``torch.nn.DataParallel`` (semseg.py:91, pcdseg.py:141): the batch is scattered
on dim 0, BatchNorm statistics are per replica, gradients are summed onto device 0.  Clouds are independent
units, so here each rank owns its own clouds end to end (FPS, grouping, MLP -- no data-path collective) and
the only exchange is ONE all-reduce (AVG) of a flat fp32 gradient bucket (3.87 MB SSG / 6.94 MB MSG) through
``torch.distributed`` -- RCCL over xGMI with backend "nccl", gloo in the CPU tests.  With per-rank mean
losses over equal shards, AVG reproduces DataParallel's global-mean gradient; BN stays per replica, which
is exactly the reference's semantics.
"""
import os

import torch
import torch.distributed as dist


def _skip_collectives(group=None):
    """No process group, or a single rank (unless PN2_FORCE_COLLECTIVES=1: lets a 1-GPU box exercise RCCL)."""
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_world_size(group) == 1 and os.environ.get("PN2_FORCE_COLLECTIVES") != "1"


def _loaded_hip_runtime():
    """Path of the HIP runtime THIS process already runs on (torch ships its own copy next to its extension modules; opening
    another copy by bare name would give a second runtime whose events mean nothing to torch's streams)."""
    try:
        with open("/proc/self/maps") as f:
            for ln in f:
                path = ln.rsplit(" ", 1)[-1].strip()
                if "libamdhip64.so" in os.path.basename(path):
                    return path
    except OSError:
        pass
    return "libamdhip64.so"


class _ExternalEvent:
    """A HIP event whose WAIT can be captured into a hipGraph while its RECORD happens outside of it, on another stream,
    after the capture: ``hipStreamWaitEvent(stream, event, hipEventWaitExternal)`` under capture becomes an event-wait node
    that, at every replay, waits for whatever was recorded on the event most recently (tools/exp/ext_event.py checks
    exactly that on gfx950 / ROCm 7).  ``torch.cuda.Event(external=True)`` is the same thing but refuses to work on ROCm
    builds ("External events are disallowed in rocm"), so this goes to the HIP runtime torch itself is linked against."""
    _hip = None

    def __init__(self, device):
        import ctypes
        if _ExternalEvent._hip is None:
            _ExternalEvent._hip = ctypes.CDLL(_loaded_hip_runtime())
        self._ct = ctypes
        self._ev = ctypes.c_void_p()
        with torch.cuda.device(device):         # an event belongs to the device that is current when it is created
            self._check(self._hip.hipEventCreateWithFlags(ctypes.byref(self._ev), 0x2), "hipEventCreateWithFlags")   # hipEventDisableTiming

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed with HIP error %d" % (what, rc))

    def record(self, stream):
        self._check(self._hip.hipEventRecord(self._ev, self._ct.c_void_p(stream.cuda_stream)), "hipEventRecord")

    def wait(self, stream):
        external = 1 if torch.cuda.is_current_stream_capturing() else 0          # hipEventWaitExternal
        self._check(self._hip.hipStreamWaitEvent(self._ct.c_void_p(stream.cuda_stream), self._ev, external), "hipStreamWaitEvent")

    def __del__(self):
        try:
            if self._ev:
                self._hip.hipEventDestroy(self._ev)
        except Exception:
            pass


class FlatGradBucket:
    """All parameter gradients as views into one contiguous fp32 buffer.

    ``p.grad`` of every parameter aliases a slice of ``self.flat``, so backward accumulates straight into
    the bucket, ``zero()`` is one memset and ``all_reduce()`` is one collective: no per-tensor launches,
    no flatten/unflatten copies.
    """

    def __init__(self, module, direct=False):
        """``direct=True`` additionally lets the HIP backward add weight / BatchNorm gradients straight into the bucket
        (pointnet_util.set_direct_grad_accumulation): no per-layer gradient tensors, no per-parameter add kernels."""
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n
        if direct:
            from . import pointnet_util
            pointnet_util.set_direct_grad_accumulation(True)
        self.comm = None          # dedicated stream of the collective (use_comm_stream)
        self.reduced = None       # event: the last all-reduce has finished

    def use_comm_stream(self):
        """Issue the all-reduce on a stream of its own.  The step that follows only has to wait where it first touches the
        bucket (``wait_reduced()`` in front of ``zero()`` / of the optimizer's read): captured into a hipGraph that is an
        event-wait node on the main branch, so the geometry-prefetch branch of the next replay (FPS, ball query, 3-NN: no
        gradient, no parameter) runs while the collective is still on the wire."""
        if self.flat.is_cuda and self.comm is None:
            self.comm = torch.cuda.Stream(device=self.flat.device)
            self.reduced = _ExternalEvent(self.flat.device)
            self.reduced.record(self.comm)
        return self

    def wait_reduced(self):
        """The calling stream waits for the last all-reduce (no-op without a comm stream)."""
        if self.reduced is not None:
            self.reduced.wait(torch.cuda.current_stream(self.flat.device))

    @property
    def nbytes(self):
        return self.flat.numel() * 4

    def zero(self):
        self.flat.zero_()

    def all_reduce(self, group=None):
        """Average the bucket over ranks (no-op for world size 1 / uninitialised process group)."""
        if _skip_collectives(group):
            return None
        self._all_reduce(group)
        return self.flat

    def all_reduce_timed(self, group=None):
        """``all_reduce`` bracketed by two timing events ON THE STREAM THE COLLECTIVE RUNS ON (the comm stream's are recorded
        after it has caught up with the step, so they time the collective, not the wait for the backward pass).  Returns
        (start, end), or None when there is no collective to time / no GPU."""
        if _skip_collectives(group):
            return None
        if not self.flat.is_cuda:
            self._all_reduce(group)
            return None
        marks = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        self._all_reduce(group, marks)
        return marks

    def _all_reduce(self, group, marks=None):
        def collective():
            if marks:
                marks[0].record()
            if dist.get_backend(group) == "gloo":          # gloo has no AVG
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
                self.flat.div_(dist.get_world_size(group))
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=group)
            if marks:
                marks[1].record()
        if self.comm is None:
            collective()
            return
        self.comm.wait_stream(torch.cuda.current_stream(self.flat.device))     # the step's gradients are complete
        with torch.cuda.stream(self.comm):
            collective()
            self.reduced.record(self.comm)


def verify_comm_stream(bucket, group=None):
    """Does the comm-stream all-reduce of ``bucket`` give the step-stream result on THIS process group?  Fills the bucket with
    rank-dependent values, reduces it once on the calling stream and once through a temporary comm stream + external event
    (the exact calls the timed steps make: ``all_reduce`` then ``wait_reduced``), compares bit for bit, and agrees on the verdict
    across ranks (MIN).  The bucket is zeroed afterwards.  False on any failure -- the caller then stays on the step stream."""
    if _skip_collectives(group) or not bucket.flat.is_cuda:
        return None
    dev = bucket.flat.device
    rank = dist.get_rank(group)
    n = bucket.flat.numel()
    pattern = ((torch.arange(n, device=dev, dtype=torch.float32) % 1021.0) - 510.0) * (1.0 + rank) / 1024.0
    ok = True
    saved = (bucket.comm, bucket.reduced)
    try:
        bucket.flat.copy_(pattern)
        bucket.comm, bucket.reduced = None, None
        bucket._all_reduce(group)
        want = bucket.flat.clone()
        bucket.flat.copy_(pattern)
        bucket.use_comm_stream()
        bucket._all_reduce(group)
        bucket.wait_reduced()
        ok = bool(torch.equal(bucket.flat, want))
    except Exception:
        ok = False
    finally:
        torch.cuda.synchronize(dev)
        bucket.comm, bucket.reduced = saved
        bucket.flat.zero_()
    verdict = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
    dist.all_reduce(verdict, op=dist.ReduceOp.MIN, group=group)
    return bool(verdict.item())


def broadcast_module(module, src=0, group=None):
    """Rank ``src``'s parameters and buffers to every rank (DataParallel's per-forward replicate, done once)."""
    if _skip_collectives(group):
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
    from . import pointnet_util
    pointnet_util.bump_param_generation()


def shard_range(global_units, rank, world):
    """Contiguous shard [lo, hi) of ``global_units`` independent clouds for ``rank`` (equal shards required)."""
    if global_units % world:
        raise ValueError("global batch %d does not split evenly over %d ranks" % (global_units, world))
    per = global_units // world
    return rank * per, (rank + 1) * per
