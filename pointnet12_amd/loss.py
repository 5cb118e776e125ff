"""The loss either side of the hot path, on the HIP library.

``nll_loss(log_probs, target)`` is a drop-in for ``torch.nn.functional.nll_loss`` as the reference calls it:
``F.nll_loss(pred, target)`` on ``[B*N, C]`` log-probabilities (semseg.py:143, mean over all points) and
the class-weighted mean of pcdseg.py:179.  ATen's kernel for this reduction runs in a single workgroup
(66 us forward + 37 us backward at 65 536 rows, fully exposed between the forward and the backward pass);
``pn2_nll_loss_fwd`` spreads it over the chip with fp64 partials combined in a fixed order.
"""
import torch

from . import _lib

_p = _lib.ptr


class _NllLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logp, target, weight, ignore_index):
        lib, st = _lib.load(), _lib.stream()
        R, C = logp.shape
        from .pointnet_util import _zeros_small                         # the zero arena: no fill launch of its own
        ws = _zeros_small(int(lib.pn2_nll_loss_workspace_bytes(R)), logp.device)
        res = torch.empty(2, device=logp.device, dtype=torch.float32)          # loss, sum of weights
        _lib.check(lib.pn2_nll_loss_fwd(_p(logp), C, _p(target), _p(weight), R, C, ignore_index, _p(ws), res.data_ptr(),
                                        res.data_ptr() + 4, st), "pn2_nll_loss_fwd")
        ctx.save_for_backward(target, weight, res)
        ctx.meta = (R, C, ignore_index)
        return res[0]

    @staticmethod
    def backward(ctx, grad):
        lib, st = _lib.load(), _lib.stream()
        target, weight, res = ctx.saved_tensors
        R, C, ignore_index = ctx.meta
        grad = grad.contiguous().float()
        dlogp = torch.empty(R, C, device=target.device, dtype=torch.float32)
        _lib.check(lib.pn2_nll_loss_bwd(_p(target), _p(weight), R, C, ignore_index, _p(grad), res.data_ptr() + 4, _p(dlogp),
                                        C, st), "pn2_nll_loss_bwd")
        return dlogp, None, None, None


def nll_loss(log_probs, target, weight=None, ignore_index=-100):
    """``F.nll_loss(log_probs, target, weight, ignore_index=ignore_index)`` with reduction "mean".

    log_probs ``[R, C]`` float32 on the GPU (``[B, N, C]`` is flattened as the reference's ``view(-1, C)``
    does), target int64 ``[R]``.  There is no CPU path: tensors must live on the HIP device."""
    if log_probs.dim() > 2:
        log_probs = log_probs.reshape(-1, log_probs.shape[-1])
    if not log_probs.is_cuda:
        raise _lib.Pn2Error("nll_loss: log_probs must be a GPU tensor (the HIP library is the only implementation)")
    if log_probs.dtype != torch.float32:
        raise TypeError("nll_loss: float32 log-probabilities expected, got %s" % log_probs.dtype)
    target = target.reshape(-1)
    if target.dtype != torch.int64:
        target = target.long()
    if target.shape[0] != log_probs.shape[0]:
        raise ValueError("nll_loss: %d rows of log-probabilities, %d targets" % (log_probs.shape[0], target.shape[0]))
    if target.device != log_probs.device:
        target = target.to(log_probs.device)
    if weight is not None:
        weight = weight.to(device=log_probs.device, dtype=torch.float32).contiguous()
        if weight.numel() != log_probs.shape[1]:
            raise ValueError("nll_loss: weight must have one entry per class")
    return _NllLoss.apply(log_probs.contiguous(), target.contiguous(), weight, int(ignore_index))
