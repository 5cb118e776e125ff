"""The optimiser step on the HIP library.

``Adam`` is a drop-in for ``torch.optim.Adam`` as the reference builds it (semseg.py:106-111, pcdseg.py:133-138:
``lr``, ``betas=(0.9, 0.999)``, ``eps=1e-08``, ``weight_decay``; amsgrad off).  It is a ``torch.optim.Optimizer``, so
``torch.optim.lr_scheduler.StepLR(optimizer, 20, 0.5)`` (semseg.py:113), ``for g in optimizer.param_groups:
g['lr'] = lr`` (pcdseg.py:162-163), ``zero_grad()`` and ``state_dict()`` work as with the original (the state it
saves has torch.optim.Adam's layout).

What differs is the memory layout: the parameters of a group are re-pointed into ONE flat fp32 buffer, their
gradients into another (``parallel.FlatGradBucket`` -- the buffer the data-parallel all-reduce already uses), and
``exp_avg`` / ``exp_avg_sq`` are flat twins, so ``step()`` is a single ``pn2_adam_step`` launch over 28 B/element
instead of ~10 foreach launches over ~150 tensors, and ``zero_grad()`` is one fill (or free: ``fused_zero_grad``).

Construct it BEFORE capturing a step into a hipGraph (graph.GraphedStep): a captured launch holds the parameter
addresses it saw, and construction moves the parameters into the flat buffer.

One semantic difference, by construction: a parameter whose gradient was never written still sees a zero gradient
(torch.optim.Adam skips ``grad is None`` parameters).  Every parameter of the reference's networks receives a
gradient in every step.
"""
import torch

from . import _lib
from . import pointnet_util
from .parallel import FlatGradBucket

_p = _lib.ptr


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, *,
                 bucket=None, fused_zero_grad=False, device_step=False):
        """``bucket``: an existing FlatGradBucket over the same parameters in the same order (single group) to share
        its gradient buffer.  ``fused_zero_grad``: ``step()`` also clears the gradients (the next ``zero_grad()``
        becomes a no-op).  ``device_step``: the step count and the learning rate live in device memory, so a
        captured ``step()`` can be replayed from a hipGraph; call ``sync_lr()`` after changing ``param_groups``."""
        if amsgrad:
            raise NotImplementedError("amsgrad is not used by the reference and not implemented")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= weight_decay:
            raise ValueError("lr, eps and weight_decay must be non-negative")
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("betas must lie in [0, 1)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.fused_zero_grad = bool(fused_zero_grad)
        self.device_step = bool(device_step)
        self._grads_clear = False
        self._flat = []
        if bucket is not None and len(self.param_groups) != 1:
            raise ValueError("a shared gradient bucket needs a single parameter group")
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.requires_grad]
            if not ps:
                raise ValueError("parameter group without trainable parameters")
            dev = ps[0].device
            if dev.type != "cuda":
                raise _lib.Pn2Error("optim.Adam: parameters must live on the GPU (the HIP library is the only "
                                    "implementation)")
            if any(p.dtype != torch.float32 or p.device != dev for p in ps):
                raise TypeError("optim.Adam: float32 parameters on one device expected")
            total = sum(p.numel() for p in ps)
            flat_p = torch.empty(total, device=dev, dtype=torch.float32)
            off = 0
            for p in ps:                                     # re-point the parameters into the flat buffer
                n = p.numel()
                view = flat_p[off:off + n].view(p.shape)
                view.copy_(p.data)
                p.data = view
                off += n
            if bucket is not None:
                if [id(p) for p in bucket.params] != [id(p) for p in ps]:
                    raise ValueError("bucket and optimizer must hold the same parameters in the same order")
                flat_g = bucket.flat
                self._bucket = bucket
            else:
                flat_g = _GradViews(ps).flat
            flat_m, flat_v = torch.zeros_like(flat_p), torch.zeros_like(flat_p)
            off = 0
            for p in ps:                                     # torch.optim.Adam's per-parameter state, as views
                n = p.numel()
                self.state[p] = {"step": torch.tensor(0.0), "exp_avg": flat_m[off:off + n].view(p.shape),
                                 "exp_avg_sq": flat_v[off:off + n].view(p.shape)}
                off += n
            rec = {"params": ps, "p": flat_p, "g": flat_g, "m": flat_m, "v": flat_v, "t": 0, "lr_dev": None,
                   "step_dev": None}
            if self.device_step:
                rec["lr_dev"] = torch.full((1,), float(group["lr"]), device=dev, dtype=torch.float32)
                rec["step_dev"] = torch.zeros(2, device=dev, dtype=torch.int64)
            self._flat.append(rec)

    def sync_lr(self):
        """Copy every group's ``lr`` to its device cell (device_step mode; call outside graph capture)."""
        for group, rec in zip(self.param_groups, self._flat):
            if rec["lr_dev"] is not None:
                rec["lr_dev"].fill_(float(group["lr"]))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib, st = _lib.load(), _lib.stream()
        if getattr(self, "_bucket", None) is not None:
            self._bucket.wait_reduced()               # an all-reduce on the bucket's comm stream must have landed
        if torch.cuda.is_current_stream_capturing() and any(rec["step_dev"] is None for rec in self._flat):
            # a captured launch would bake the host step count and learning rate in as kernel-argument constants:
            # every replay would then repeat the SAME bias correction and lr, silently
            raise _lib.Pn2Error("optim.Adam.step() under stream capture needs device_step=True (step count and lr "
                                "in device memory); with host-side values every graph replay would reuse this step's")
        for group, rec in zip(self.param_groups, self._flat):
            for p in rec["params"]:
                if p.grad is None or p.grad.data_ptr() < rec["g"].data_ptr() or \
                        p.grad.data_ptr() >= rec["g"].data_ptr() + rec["g"].numel() * 4:
                    raise _lib.Pn2Error("optim.Adam: a parameter's .grad no longer aliases the flat gradient buffer "
                                        "(use zero_grad(), not `p.grad = None`)")
            rec["t"] += 1
            b1, b2 = group["betas"]
            _lib.check(lib.pn2_adam_step(_p(rec["p"]), _p(rec["g"]), _p(rec["m"]), _p(rec["v"]), rec["p"].numel(),
                                         float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                         float(group["weight_decay"]), rec["t"], _p(rec["lr_dev"]), _p(rec["step_dev"]),
                                         int(self.fused_zero_grad), st), "pn2_adam_step")
        self._grads_clear = self.fused_zero_grad
        pointnet_util.bump_param_generation()       # parameters written through raw pointers: eval-mode folds are stale
        return loss

    def zero_grad(self, set_to_none=False):
        """One fill per group (the flat buffers stay attached: ``set_to_none`` is ignored)."""
        if self._grads_clear:                                # the last step() already cleared them; only once
            self._grads_clear = False
            return
        for rec in self._flat:
            rec["g"].zero_()

    def steps_taken(self):
        """Host view of the step count per group (device_step mode reads the device cell: synchronises)."""
        return [int(rec["step_dev"][0]) if rec["step_dev"] is not None else rec["t"] for rec in self._flat]

    def state_dict(self):
        for rec, taken in zip(self._flat, self.steps_taken()):
            for p in rec["params"]:
                self.state[p]["step"] = torch.tensor(float(taken))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """Accepts a torch.optim.Adam (or this class's) state dict; moments are copied into the flat buffers."""
        views = {p: (self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"]) for rec in self._flat for p in rec["params"]}
        super().load_state_dict(state_dict)
        for rec in self._flat:
            t = 0
            for p in rec["params"]:
                s = self.state.get(p, {})
                m, v = views[p]
                if "exp_avg" in s:
                    m.copy_(s["exp_avg"])
                    v.copy_(s["exp_avg_sq"])
                    t = int(float(s.get("step", 0)))
                self.state[p] = {"step": torch.tensor(float(t)), "exp_avg": m, "exp_avg_sq": v}
            rec["t"] = t
            if rec["step_dev"] is not None:
                rec["step_dev"][0] = t
        self.sync_lr()


class _GradViews(FlatGradBucket):
    """A FlatGradBucket over an explicit parameter list."""

    def __init__(self, params):
        self.params = list(params)
        dev = self.params[0].device
        self.flat = torch.zeros(sum(p.numel() for p in self.params), device=dev, dtype=torch.float32)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n
