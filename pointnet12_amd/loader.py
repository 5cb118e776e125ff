"""Per-batch cloud preparation on the HIP library.

The reference prepares every training item on the host (``SemKITTI_Loader.__getitem__``,
data_utils/SemKITTI_Loader.py:93-113): ``pcd_normalize`` (:23-30), ``pcd_jitter`` (:17-21, training only), then
``np.random.choice(length, npoints, replace=True)`` and two fancy-index gathers (:110-113); the collated batch is
copied to the GPU and transposed (semseg.py:131).  Here the raw scans stay resident in HBM (``ScanStore``: all of
SemanticKITTI's training split is ~25 GB, 288 GB are available) and one ``pn2_prepare_clouds`` launch writes the
``[B, N, 4]`` batch and its ``[B, N]`` labels; only the random draws cross the host boundary.

Two sources for the draws:
  * ``rng="numpy"`` (default): numpy's global generator in the reference's order -- per cloud ``randn(M, 4)`` (when
    training) then ``choice(M, npoints)`` -- so ``np.random.seed(s)`` reproduces the reference's batches bit for
    bit (a ``num_workers=0`` loader; worker processes reseed numpy in the reference as well).
  * ``rng=torch.Generator(device)``: draws on the device (same distributions, different stream): nothing but the
    scan numbers crosses PCIe.
"""
import numpy as np
import torch

from . import _lib

_p = _lib.ptr


class ScanStore:
    """Raw scans ``[M_i, 4]`` fp32 (x, y, z, intensity -- the .bin rows of kitti_utils.py:200) and their int32
    per-point classes, uploaded once and kept back to back in HBM."""

    def __init__(self, scans, labels=None, device="cuda"):
        if not scans:
            raise ValueError("ScanStore: no scans")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.Pn2Error("ScanStore: the HIP device is the only implementation")
        counts = [int(s.shape[0]) for s in scans]
        for s in scans:
            if s.ndim != 2 or s.shape[1] != 4:
                raise ValueError("ScanStore: scans must be [M, 4] (x, y, z, intensity)")
        self.row_count = torch.tensor(counts, dtype=torch.int64)
        self.row_begin = torch.cumsum(self.row_count, 0) - self.row_count
        self.raw = torch.from_numpy(np.ascontiguousarray(np.concatenate(scans, 0), np.float32)).to(self.device)
        self.label = None
        if labels is not None:
            if [int(l.shape[0]) for l in labels] != counts:
                raise ValueError("Scan and Label don't contain same number of points")     # kitti_utils.py:211
            self.label = torch.from_numpy(np.ascontiguousarray(np.concatenate(labels, 0), np.int32)).to(self.device)
        self._begin_dev = self.row_begin.to(self.device)
        self._count_dev = self.row_count.to(self.device)

    def __len__(self):
        return self.row_count.numel()


def pcd_jitter_noise(M, C=4, sigma=0.01, clip=0.05):
    """The clipped jitter rows of ``pcd_jitter`` (SemKITTI_Loader.py:17-19), from numpy's global generator."""
    return np.clip(sigma * np.random.randn(M, C), -1 * clip, clip).astype(np.float32)


def prepare_batch(store, scan_ids, npoints, train=True, rng="numpy", sigma=0.01, clip=0.05, out=None):
    """Batch of ``len(scan_ids)`` clouds: ``(points [B, npoints, 4] fp32, labels [B, npoints] int64 | None)`` on the
    device.  ``points.transpose(2, 1)`` is the ``[B, 4, N]`` tensor semseg.py:131 feeds the network.  ``out``: a
    ``(points, labels)`` pair of contiguous tensors to write into (the static inputs of a captured step)."""
    lib, st = _lib.load(), _lib.stream()
    ids = torch.as_tensor(scan_ids, dtype=torch.int64)
    B = ids.numel()
    if B == 0:
        raise ValueError("prepare_batch: empty batch")
    counts = store.row_count[ids]
    dev = store.device
    noise = noise_begin = None
    if rng == "numpy":
        rows, picks = [], []
        for m in counts.tolist():
            if train:
                rows.append(pcd_jitter_noise(m, 4, sigma, clip))
            picks.append(np.random.choice(m, npoints, replace=True))
        choice = torch.from_numpy(np.stack(picks).astype(np.int64)).to(dev)
        if train:
            noise = torch.from_numpy(np.concatenate(rows, 0)).to(dev)
            noise_begin = (torch.cumsum(counts, 0) - counts).to(dev)
    elif isinstance(rng, torch.Generator):
        cnt = counts.to(dev)
        u = torch.rand(B, npoints, device=dev, dtype=torch.float64, generator=rng)
        choice = torch.minimum((u * cnt[:, None]).long(), cnt[:, None] - 1)
        if train:
            total = int(counts.sum())
            noise = torch.randn(total, 4, device=dev, generator=rng).mul_(sigma).clamp_(-clip, clip)
            noise_begin = (torch.cumsum(counts, 0) - counts).to(dev)
    else:
        raise ValueError('prepare_batch: rng must be "numpy" or a device torch.Generator')
    ids_dev = ids.to(dev)
    begin, count = store._begin_dev[ids_dev], store._count_dev[ids_dev]
    if out is not None:
        points, labels = out
        if points.shape != (B, npoints, 4) or points.dtype != torch.float32 or not points.is_contiguous() or \
                (labels is not None and (labels.shape != (B, npoints) or labels.dtype != torch.int64
                                         or not labels.is_contiguous())):
            raise ValueError("prepare_batch: out must be contiguous ([B, npoints, 4] float32, [B, npoints] int64)")
        if labels is not None and store.label is None:
            raise ValueError("prepare_batch: the store holds no labels")
    else:
        points = torch.empty(B, npoints, 4, device=dev, dtype=torch.float32)
        labels = torch.empty(B, npoints, device=dev, dtype=torch.int64) if store.label is not None else None
    _lib.check(lib.pn2_prepare_clouds(_p(store.raw), _p(begin), _p(count), _p(store.label), _p(noise), _p(noise_begin),
                                      _p(choice), B, npoints, _p(points), _p(labels), None, st), "pn2_prepare_clouds")
    if out is not None:
        from . import pointnet_util as _U
        _U.bump_data_generation()             # a static buffer was refilled through a raw pointer: views of it are stale
    return points, labels
