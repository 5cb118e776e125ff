"""torch-CPU restatement of the three PointNet++ modules and the two benchmark networks.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__``).  Index tensors come from the scalar C
restatement (``oracle.geometry``); rows are gathered with ``index_select`` on flat tables and
the shared MLP runs functionally (``F.conv2d``/``F.conv1d`` + ``F.batch_norm`` + relu) in the
reference's channel-first layout (model/pointnet_util.py:194-199, :251-256, :309-312).
``tools/make_golden.py`` asserts outputs and gradients against the reference itself.

Parameter containers are real ``nn.Conv2d``/``nn.BatchNorm2d`` (``Conv1d``/``BatchNorm1d`` for
FP) under the reference's attribute names, created in the reference's order, so a reference
``state_dict`` loads unchanged and seeded initialisation draws the same values.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import geometry as G
from .aten_geometry import AtenGeometry as A

# "c": index tensors from the scalar C restatement (the parity checker's default).  "aten": the reference's own
# operator sequence (oracle/aten_geometry.py) -- same indices bit for bit, the reference's cost structure; used by
# bench.py's cpu_baseline leg so that the timed CPU step is the reference's step, not a faster algorithm.
_GEOMETRY = "c"


def set_geometry(mode):
    global _GEOMETRY
    if mode not in ("c", "aten"):
        raise ValueError(mode)
    _GEOMETRY = mode


# tests/test_parity_stages_gpu.py: a list that receives (arg-max index, max value), each [B,C,S], of every pooled MLP stack
# (the index torch.max(x, 2) of pointnet_util.py:199 / :256 routes the gradient to); None = off
POOL_ARGMAX = None


# tests/test_parity_stages_gpu.py, decisions: the DISCRETE choices of a shared MLP -- every layer's ReLU mask (relu(bn(conv)) of
# pointnet_util.py:197 / :255 / :312) and the pooled arg-max (:199 / :256).  None = off;
#   {"record": []}             every stack appends {"masks": [bool [B,C,K,S] / [B,C,N] per layer], "argmax": int64 [B,C,S] or None}
#   {"force": [...], "pos": 0} every stack takes the next entry and evaluates  bn(.) * mask  instead of relu(bn(.)) and a gather at
#                              the given index instead of torch.max: the same function wherever the decisions agree, and the
#                              smooth function "with those decisions" where they do not (what an evaluation that flipped a
#                              decision within rounding of its threshold computes -- gradients included).
DECISIONS = None
_LAST_STACK = None


def _next_stack():
    global _LAST_STACK
    dec = DECISIONS
    if dec is None:
        _LAST_STACK = None
    elif "force" in dec:
        _LAST_STACK = dec["force"][dec["pos"]]
        dec["pos"] += 1
    else:
        _LAST_STACK = {"masks": [], "argmax": None}
        dec["record"].append(_LAST_STACK)
    return _LAST_STACK


def _relu_decided(z, stack, layer):
    if stack is None:
        return F.relu(z)
    if "force" in DECISIONS:
        return z * stack["masks"][layer].to(z.dtype)
    stack["masks"].append((z > 0).detach())
    return F.relu(z)


def _pool(y):
    """torch.max(new_points, 2)[0] of pointnet_util.py:199 / :256 on [B,C,K,S]."""
    stack = _LAST_STACK if DECISIONS is not None else None
    if stack is not None and "force" in DECISIONS:
        return y.gather(2, stack["argmax"].unsqueeze(2)).squeeze(2)
    m = y.max(dim=2)
    if stack is not None:
        stack["argmax"] = m[1].detach()
    if POOL_ARGMAX is not None:
        POOL_ARGMAX.append((m[1].detach(), m[0].detach()))
    return m[0]


def _fps(xyz_r, S, start):
    if _GEOMETRY == "aten":
        return A.fps(xyz_r.detach(), S, start)
    return torch.from_numpy(G.farthest_point_sample(xyz_r.detach().numpy(), S, start.numpy()))


def _ball(radius, K, xyz_r, new_xyz):
    if _GEOMETRY == "aten":
        return A.ball(radius, K, xyz_r.detach(), new_xyz.detach())
    return torch.from_numpy(G.query_ball_point(radius, K, xyz_r.detach().numpy(), new_xyz.detach().numpy()))


def _rows(table, idx):
    """table [B,N,C], idx [B,...] -> [B,...,C]; differentiable w.r.t. table."""
    if _GEOMETRY == "aten":
        return A.gather(table, idx)
    B, N, C = table.shape
    return _take_rows(table.reshape(B * N, C), _flat(idx, N)).view(tuple(idx.shape) + (C,))


def _shared_mlp(x, convs, bns, training):
    """relu(bn(conv1x1(.))) per layer, channel-first [B,C,K,S] (SA) or [B,C,N] (FP).

    Kept in the reference's tensor layout on purpose: ATen's channel-first batch-norm
    reduces each channel with a cascade sum and stays ~2e-6 from an fp64 evaluation, while
    the same statistics over a [P, C] row matrix drift to 1e-5 (measured; see HISTORY.md section 1).
    """
    y = x
    stack = _next_stack()
    for layer, (conv, bn) in enumerate(zip(convs, bns)):
        y = F.conv2d(y, conv.weight, conv.bias) if y.dim() == 4 else F.conv1d(y, conv.weight, conv.bias)
        if training and bn.track_running_stats:
            bn.num_batches_tracked += 1
        y = F.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias,
                         training, bn.momentum, bn.eps)
        y = _relu_decided(y, stack, layer)
    return y


def _take_rows(table, flat_idx):
    """table [B*N, C], flat_idx int64 [M] -> [M, C]; differentiable w.r.t. table."""
    return table.index_select(0, flat_idx)


def _flat(idx, n):
    """[B, ...] per-cloud indices -> flat indices into a [B*n, C] table."""
    b = idx.shape[0]
    off = (torch.arange(b, dtype=torch.long) * n).view([b] + [1] * (idx.dim() - 1))
    return (idx + off).reshape(-1)


def draw_start(batch, n):
    """The FPS start draw of pointnet_util.py:75: CPU default generator, one call per FPS."""
    return torch.randint(0, n, (batch,), dtype=torch.long)


class RefSetAbstraction(nn.Module):
    """Restates PointNetSetAbstraction (model/pointnet_util.py:160-201)."""

    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all):
        super().__init__()
        self.npoint, self.radius, self.nsample, self.group_all = npoint, radius, nsample, group_all
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        c = in_channel
        for o in mlp:
            self.mlp_convs.append(nn.Conv2d(c, o, 1))
            self.mlp_bns.append(nn.BatchNorm2d(o))
            c = o

    def forward(self, xyz, points, start=None):
        xyz_r = xyz.permute(0, 2, 1).contiguous()                      # [B,N,3]   (:184)
        pts_r = None if points is None else points.permute(0, 2, 1).contiguous()
        B, N, _ = xyz_r.shape
        if self.group_all:                                             # :140-157
            new_xyz = torch.zeros(B, 1, 3)
            rows = xyz_r if pts_r is None else torch.cat([xyz_r, pts_r], -1)
            S, K = 1, N
        else:                                                          # :110-137
            S, K = self.npoint, self.nsample
            if start is None:
                start = draw_start(B, N)
            fidx = _fps(xyz_r, S, start)
            new_xyz = _rows(xyz_r, fidx)
            gidx = _ball(self.radius, K, xyz_r, new_xyz)
            if int(gidx.max()) >= N:
                raise IndexError("empty ball: index N reaches index_points (pointnet_util.py:127)")
            gx = _rows(xyz_r, gidx) - new_xyz.view(B, S, 1, 3)
            if pts_r is None:
                rows = gx
            else:
                rows = torch.cat([gx, _rows(pts_r, gidx)], -1)         # xyz first (:131)
        y = _shared_mlp(rows.view(B, S, K, -1).permute(0, 3, 2, 1), self.mlp_convs, self.mlp_bns,
                        self.training)                                  # [B,C,K,S]  (:194-197)
        return new_xyz.permute(0, 2, 1), _pool(y)                      # :199


class RefSetAbstractionMsg(nn.Module):
    """Restates PointNetSetAbstractionMsg (model/pointnet_util.py:204-261)."""

    def __init__(self, npoint, radius_list, nsample_list, in_channel, mlp_list):
        super().__init__()
        self.npoint, self.radius_list, self.nsample_list = npoint, radius_list, nsample_list
        self.conv_blocks = nn.ModuleList()
        self.bn_blocks = nn.ModuleList()
        for mlp in mlp_list:
            convs, bns = nn.ModuleList(), nn.ModuleList()
            c = in_channel + 3                                         # :215
            for o in mlp:
                convs.append(nn.Conv2d(c, o, 1))
                bns.append(nn.BatchNorm2d(o))
                c = o
            self.conv_blocks.append(convs)
            self.bn_blocks.append(bns)

    def forward(self, xyz, points, start=None):
        xyz_r = xyz.permute(0, 2, 1).contiguous()
        pts_r = None if points is None else points.permute(0, 2, 1).contiguous()
        B, N, _ = xyz_r.shape
        S = self.npoint
        if start is None:
            start = draw_start(B, N)
        fidx = _fps(xyz_r, S, start)
        new_xyz = _rows(xyz_r, fidx)                                                    # :238
        outs = []
        for radius, K, convs, bns in zip(self.radius_list, self.nsample_list, self.conv_blocks, self.bn_blocks):
            gidx = _ball(radius, K, xyz_r, new_xyz)
            if int(gidx.max()) >= N:
                raise IndexError("empty ball: index N reaches index_points (pointnet_util.py:243)")
            gx = _rows(xyz_r, gidx) - new_xyz.view(B, S, 1, 3)
            if pts_r is None:
                rows = gx
            else:
                rows = torch.cat([_rows(pts_r, gidx), gx], -1)         # features first (:247)
            y = _shared_mlp(rows.permute(0, 3, 2, 1), convs, bns, self.training)      # :251-255
            outs.append(_pool(y))                                      # :256
        return new_xyz.permute(0, 2, 1), torch.cat(outs, 1)            # :260


class RefFeaturePropagation(nn.Module):
    """Restates PointNetFeaturePropagation (model/pointnet_util.py:264-313)."""

    def __init__(self, in_channel, mlp):
        super().__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        c = in_channel
        for o in mlp:
            self.mlp_convs.append(nn.Conv1d(c, o, 1))
            self.mlp_bns.append(nn.BatchNorm1d(o))
            c = o

    def forward(self, xyz1, xyz2, points1, points2):
        x1 = xyz1.permute(0, 2, 1).contiguous()
        x2 = xyz2.permute(0, 2, 1).contiguous()
        p2 = points2.permute(0, 2, 1).contiguous()
        B, N, _ = x1.shape
        S = x2.shape[1]
        if S == 1:                                                     # :292-293
            interp = p2.expand(B, N, p2.shape[-1])
        elif _GEOMETRY == "aten":
            interp, _ = A.three_nn_interp(x1.detach(), x2.detach(), p2)                 # :295-301
        else:
            idx, dist = G.three_nn(x1.detach().numpy(), x2.detach().numpy())           # :295-297
            w = torch.from_numpy(G.three_weights(dist))                                 # :298-300
            nb = _take_rows(p2.reshape(B * S, -1), _flat(torch.from_numpy(idx), S)).view(B, N, 3, -1)
            t = nb * w.view(B, N, 3, 1)
            interp = (t[:, :, 0] + t[:, :, 1]) + t[:, :, 2]                            # :301
        if points1 is not None:
            rows = torch.cat([points1.permute(0, 2, 1), interp], -1)   # points1 first (:305)
        else:
            rows = interp
        return _shared_mlp(rows.permute(0, 2, 1), self.mlp_convs, self.mlp_bns, self.training)  # :308-312


class _SegHead(nn.Module):
    """conv1/bn1/drop1/conv2 + log_softmax of model/pointnet2.py:154-157,172-175."""

    def _make_head(self, num_classes, dropout):
        # called last by the subclasses: registration order sa*, fp*, head (as the reference)
        self.conv1 = nn.Conv1d(128, 128, 1)
        self.bn1 = nn.BatchNorm1d(128)
        self.drop1 = nn.Dropout(dropout)
        self.conv2 = nn.Conv1d(128, num_classes, 1)

    def head(self, feat):
        x = self.drop1(_relu_decided(self.bn1(self.conv1(feat)), _next_stack(), 0))
        x = F.log_softmax(self.conv2(x), dim=1)
        return x.permute(0, 2, 1)


class RefSSGSemSeg(_SegHead):
    """Restates PointNet2SemSeg (model/pointnet2.py:141-176): 4 single-scale SA + 4 FP."""

    def __init__(self, num_classes, feature_dims=3, dropout=0.5):
        super().__init__()
        d = feature_dims
        self.feature_dims = d
        self.sa1 = RefSetAbstraction(1024, 0.1, 32, d + 3, [32, 32, 64], False)
        self.sa2 = RefSetAbstraction(256, 0.2, 32, 64 + 3, [64, 64, 128], False)
        self.sa3 = RefSetAbstraction(64, 0.4, 32, 128 + 3, [128, 128, 256], False)
        self.sa4 = RefSetAbstraction(16, 0.8, 32, 256 + 3, [256, 256, 512], False)
        self.fp4 = RefFeaturePropagation(768, [256, 256])
        self.fp3 = RefFeaturePropagation(384, [256, 256])
        self.fp2 = RefFeaturePropagation(320, [256, 128])
        self.fp1 = RefFeaturePropagation(128, [128, 128, 128])
        self._make_head(num_classes, dropout)

    def forward(self, points):
        xyz, feat = points[:, :3, :], points[:, 3:, :]
        x1, f1 = self.sa1(xyz, feat)
        x2, f2 = self.sa2(x1, f1)
        x3, f3 = self.sa3(x2, f2)
        x4, f4 = self.sa4(x3, f3)
        f3 = self.fp4(x3, x4, f3, f4)
        f2 = self.fp3(x2, x3, f2, f3)
        f1 = self.fp2(x1, x2, f1, f2)
        f0 = self.fp1(xyz, x1, None, f1)
        return self.head(f0)


class RefMSGSemSeg(_SegHead):
    """MSG-SemSeg of SURVEY.md §8(d): PointNet2PartSegMsg_one_hot (model/pointnet2.py:106-139)
    without the 16-channel one-hot label, with D extra input features."""

    def __init__(self, num_classes, feature_dims=6, dropout=0.5, npoint_scale=1):
        super().__init__()
        d = feature_dims
        self.feature_dims = d
        # npoint_scale: BASELINE.json cfg5 (dense 65 536-point scans) scales the sampled-point counts by 16
        self.sa1 = RefSetAbstractionMsg(512 * npoint_scale, [0.1, 0.2, 0.4], [32, 64, 128], d,
                                        [[32, 32, 64], [64, 64, 128], [64, 96, 128]])
        self.sa2 = RefSetAbstractionMsg(128 * npoint_scale, [0.4, 0.8], [64, 128], 128 + 128 + 64,
                                        [[128, 128, 256], [128, 196, 256]])
        self.sa3 = RefSetAbstraction(None, None, None, 512 + 3, [256, 512, 1024], True)
        self.fp3 = RefFeaturePropagation(1536, [256, 256])
        self.fp2 = RefFeaturePropagation(576, [256, 128])
        self.fp1 = RefFeaturePropagation(128 + 3 + d, [128, 128])
        self._make_head(num_classes, dropout)

    def forward(self, points):
        xyz, feat = points[:, :3, :], points[:, 3:, :]
        x1, f1 = self.sa1(xyz, feat)
        x2, f2 = self.sa2(x1, f1)
        x3, f3 = self.sa3(x2, f2)
        f2 = self.fp3(x2, x3, f2, f3)
        f1 = self.fp2(x1, x2, f1, f2)
        f0 = self.fp1(xyz, x1, torch.cat([xyz, feat], 1), f1)
        return self.head(f0)


class _ClsHead(nn.Module):
    """fc1/bn1/drop1/fc2/bn2/drop2/fc3 + log_softmax of model/pointnet2.py:29-46, :56-72."""

    def _make_cls_head(self, dropout):
        self.fc1 = nn.Linear(1024, 512)
        self.bn1 = nn.BatchNorm1d(512)
        self.drop1 = nn.Dropout(dropout)
        self.fc2 = nn.Linear(512, 256)
        self.bn2 = nn.BatchNorm1d(256)
        self.drop2 = nn.Dropout(dropout)
        self.fc3 = nn.Linear(256, 40)

    def cls_head(self, l3):
        x = l3.view(l3.shape[0], 1024)
        x = self.drop1(F.relu(self.bn1(self.fc1(x))))
        x = self.drop2(F.relu(self.bn2(self.fc2(x))))
        return F.log_softmax(self.fc3(x), -1)


class RefClsMsg(_ClsHead):
    """Restates PointNet2ClsMsg (model/pointnet2.py:7-47): returns (log_probs, l3_points)."""

    def __init__(self, dropout=0.4):
        super().__init__()
        self.sa1 = RefSetAbstractionMsg(512, [0.1, 0.2, 0.4], [16, 32, 128], 0,
                                        [[32, 32, 64], [64, 64, 128], [64, 96, 128]])
        self.sa2 = RefSetAbstractionMsg(128, [0.2, 0.4, 0.8], [32, 64, 128], 320,
                                        [[64, 64, 128], [128, 128, 256], [128, 128, 256]])
        self.sa3 = RefSetAbstraction(None, None, None, 640 + 3, [256, 512, 1024], True)
        self._make_cls_head(dropout)

    def forward(self, xyz):
        x1, f1 = self.sa1(xyz, None)
        x2, f2 = self.sa2(x1, f1)
        _, f3 = self.sa3(x2, f2)
        return self.cls_head(f3), f3


class RefClsSsg(_ClsHead):
    """Restates PointNet2ClsSsg (model/pointnet2.py:49-73)."""

    def __init__(self, dropout=0.4):
        super().__init__()
        self.sa1 = RefSetAbstraction(512, 0.2, 32, 3, [64, 64, 128], False)
        self.sa2 = RefSetAbstraction(128, 0.4, 64, 128 + 3, [128, 128, 256], False)
        self.sa3 = RefSetAbstraction(None, None, None, 256 + 3, [256, 512, 1024], True)
        self._make_cls_head(dropout)

    def forward(self, xyz):
        x1, f1 = self.sa1(xyz, None)
        x2, f2 = self.sa2(x1, f1)
        _, f3 = self.sa3(x2, f2)
        return self.cls_head(f3)


class RefPartSegSsg(_SegHead):
    """Restates PointNet2PartSegSsg (model/pointnet2.py:75-104): returns (log_probs [B,N,C], feat [B,128,N])."""

    def __init__(self, num_classes, dropout=0.5):
        super().__init__()
        self.sa1 = RefSetAbstraction(512, 0.2, 64, 3, [64, 64, 128], False)
        self.sa2 = RefSetAbstraction(128, 0.4, 64, 128 + 3, [128, 128, 256], False)
        self.sa3 = RefSetAbstraction(None, None, None, 256 + 3, [256, 512, 1024], True)
        self.fp3 = RefFeaturePropagation(1280, [256, 256])
        self.fp2 = RefFeaturePropagation(384, [256, 128])
        self.fp1 = RefFeaturePropagation(128, [128, 128, 128])
        self._make_head(num_classes, dropout)

    def forward(self, xyz):
        x1, f1 = self.sa1(xyz, None)
        x2, f2 = self.sa2(x1, f1)
        x3, f3 = self.sa3(x2, f2)
        f2 = self.fp3(x2, x3, f2, f3)
        f1 = self.fp2(x1, x2, f1, f2)
        f0 = self.fp1(xyz, x1, None, f1)
        feat = F.relu(self.bn1(self.conv1(f0)))                        # :97
        x = F.log_softmax(self.conv2(self.drop1(feat)), dim=1)
        return x.permute(0, 2, 1), feat


class RefPartSegMsgOneHot(_SegHead):
    """Restates PointNet2PartSegMsg_one_hot (model/pointnet2.py:106-139)."""

    def __init__(self, num_classes, dropout=0.5):
        super().__init__()
        self.sa1 = RefSetAbstractionMsg(512, [0.1, 0.2, 0.4], [32, 64, 128], 0 + 3,
                                        [[32, 32, 64], [64, 64, 128], [64, 96, 128]])
        self.sa2 = RefSetAbstractionMsg(128, [0.4, 0.8], [64, 128], 128 + 128 + 64,
                                        [[128, 128, 256], [128, 196, 256]])
        self.sa3 = RefSetAbstraction(None, None, None, 512 + 3, [256, 512, 1024], True)
        self.fp3 = RefFeaturePropagation(1536, [256, 256])
        self.fp2 = RefFeaturePropagation(576, [256, 128])
        self.fp1 = RefFeaturePropagation(150, [128, 128])
        self._make_head(num_classes, dropout)

    def forward(self, xyz, norm_plt, cls_label):
        B, _, N = xyz.shape
        x1, f1 = self.sa1(xyz, norm_plt)
        x2, f2 = self.sa2(x1, f1)
        x3, f3 = self.sa3(x2, f2)
        f2 = self.fp3(x2, x3, f2, f3)
        f1 = self.fp2(x1, x2, f1, f2)
        one_hot = cls_label.view(B, 16, 1).repeat(1, 1, N)                             # :129
        f0 = self.fp1(xyz, x1, torch.cat([one_hot, xyz, norm_plt], 1), f1)             # :130
        return self.head(f0)


def count_params(m):
    return int(sum(p.numel() for p in m.parameters()))


def seg_loss(log_probs, labels):
    """F.nll_loss over all B*N points (reference semseg.py:141-143)."""
    c = log_probs.shape[-1]
    return F.nll_loss(log_probs.reshape(-1, c), torch.as_tensor(labels).reshape(-1))


def numpy_state(module):
    return {k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def load_numpy_state(module, arrays):
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in arrays.items()}
    module.load_state_dict(sd)
