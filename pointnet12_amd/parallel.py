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

    @property
    def nbytes(self):
        return self.flat.numel() * 4

    def zero(self):
        self.flat.zero_()

    def all_reduce(self, group=None):
        """Average the bucket over ranks (no-op for world size 1 / uninitialised process group)."""
        if _skip_collectives(group):
            return None
        if dist.get_backend(group) == "gloo":          # gloo has no AVG
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(dist.get_world_size(group))
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=group)
        return self.flat


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
