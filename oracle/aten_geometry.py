"""The reference's geometry as the SAME SEQUENCE OF ATen OPERATORS, for the timed CPU baseline.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__``): imported by ``bench.py``'s ``cpu_baseline`` leg and by the CPU
tests that pin it, never by ``pointnet12_amd/``.

``oracle/pn2_oracle.c`` states WHAT the reference computes (scalar C, rounding order spelled out) and is the parity
checker.  It is also far faster than the reference -- no dense ``[B,S,N]`` distance matrix, no full-row sort, no
1 024-trip Python loop -- so timing it would flatter the CPU side.  BASELINE.md section 3 asks for the reference's
own cost structure on the GPU node's host cores: this module issues, per primitive, the operator sequence the
reference issues (model/pointnet_util.py:19-107, :295-301; one line of reference per line here, cited), so its
step time tracks the reference's (asserted within +-10 % in the development container, tools/make_golden.py g11)
while every index it returns is asserted bit-equal to the C restatement (tests/test_oracle_golden.py).
"""
import torch


class AtenGeometry:
    """Drop-in for the functions of ``oracle.geometry`` that ``oracle.torch_ref`` calls, on torch tensors."""

    @staticmethod
    def pair_sqdist(a, b):
        # :37-39  matmul, then the two squared norms added in place (a: [B,S,3], b: [B,N,3] -> [B,S,N])
        d = -2 * torch.matmul(a, b.permute(0, 2, 1))
        d += torch.sum(a ** 2, -1).view(a.shape[0], a.shape[1], 1)
        d += torch.sum(b ** 2, -1).view(b.shape[0], 1, b.shape[1])
        return d

    @staticmethod
    def gather(table, idx):
        # :52-59  batched advanced indexing with an expanded batch-index tensor
        lead = [table.shape[0]] + [1] * (idx.dim() - 1)
        tile = [1] + list(idx.shape[1:])
        rows = torch.arange(table.shape[0], dtype=torch.long).view(lead).repeat(tile)
        return table[rows, idx, :]

    @staticmethod
    def fps(xyz, npoint, start):
        # :73-84  npoint trips of: record, fetch centroid, distances, masked min update, argmax
        batch, n, _ = xyz.shape
        picks = torch.zeros(batch, npoint, dtype=torch.long)
        nearest = torch.ones(batch, n) * 1e10
        cur = start
        rows = torch.arange(batch, dtype=torch.long)
        for i in range(npoint):
            picks[:, i] = cur
            c = xyz[rows, cur, :].view(batch, 1, 3)
            d = torch.sum((xyz - c) ** 2, -1)
            closer = d < nearest
            nearest[closer] = d[closer]
            cur = torch.max(nearest, -1)[1]
        return picks

    @classmethod
    def ball(cls, radius, nsample, xyz, centres):
        # :98-106  dense candidate-index tensor, dense distances, mask, full-row sort, pad with the first hit
        batch, n, _ = xyz.shape
        s = centres.shape[1]
        cand = torch.arange(n, dtype=torch.long).view(1, 1, n).repeat([batch, s, 1])
        d = cls.pair_sqdist(centres, xyz)
        cand[d > radius ** 2] = n
        cand = cand.sort(dim=-1)[0][:, :, :nsample]
        first = cand[:, :, 0].view(batch, s, 1).repeat([1, 1, nsample])
        empty = cand == n
        cand[empty] = first[empty]
        return cand

    @classmethod
    def three_nn_interp(cls, xyz1, xyz2, points2):
        # :295-301  dense distances, full-row sort, three nearest, clamped reciprocal weights, weighted sum
        batch, n, _ = xyz1.shape
        d, idx = cls.pair_sqdist(xyz1, xyz2).sort(dim=-1)
        d, idx = d[:, :, :3], idx[:, :, :3]
        d[d < 1e-10] = 1e-10
        w = 1.0 / d
        w = w / torch.sum(w, dim=-1).view(batch, n, 1)
        return torch.sum(cls.gather(points2, idx) * w.view(batch, n, 3, 1), dim=2), idx
