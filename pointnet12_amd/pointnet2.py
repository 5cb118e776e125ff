"""PointNet++ networks composed from the MI355X-native modules.

Counterpart of the reference's ``model/pointnet2.py`` (the only importer of the hot path,
``model/pointnet2.py:5``): same class names, constructor arguments, attribute names and therefore
the same ``state_dict`` keys, so reference checkpoints (``module.``-prefixed, model/utils.py:22-27)
load with :func:`load_reference_state`.  ``PointNet2SemSegMsg`` is the MSG-SemSeg benchmark
network of SURVEY.md §8(d): ``PointNet2PartSegMsg_one_hot`` without the one-hot class label.

The segmentation head (Conv1d + BatchNorm1d + ReLU + Dropout + Conv1d + log_softmax,
model/pointnet2.py:154-175) reuses the HIP shared-MLP kernels for conv1/bn1/relu; the classification
head (three Linear layers on a [B,1024] vector, model/pointnet2.py:29-46) is stock PyTorch-ROCm.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .pointnet_util import (PointNetFeaturePropagation, PointNetSetAbstraction, PointNetSetAbstractionMsg,
                            conv1x1, log_softmax_rows, shared_mlp)


class _ClsHead(nn.Module):
    def _make_cls_head(self, num_classes=40):
        self.fc1 = nn.Linear(1024, 512)
        self.bn1 = nn.BatchNorm1d(512)
        self.drop1 = nn.Dropout(0.4)
        self.fc2 = nn.Linear(512, 256)
        self.bn2 = nn.BatchNorm1d(256)
        self.drop2 = nn.Dropout(0.4)
        self.fc3 = nn.Linear(256, num_classes)

    def _cls_head(self, l3_points):
        x = l3_points.reshape(l3_points.shape[0], 1024)
        x = self.drop1(F.relu(self.bn1(self.fc1(x))))
        x = self.drop2(F.relu(self.bn2(self.fc2(x))))
        return F.log_softmax(self.fc3(x), -1)


class _SegHead(nn.Module):
    def _make_seg_head(self, num_classes):
        self.conv1 = nn.Conv1d(128, 128, 1)
        self.bn1 = nn.BatchNorm1d(128)
        self.drop1 = nn.Dropout(0.5)
        self.conv2 = nn.Conv1d(128, num_classes, 1)

    def _seg_head(self, l0_points):
        """conv1 + bn1 + relu + drop1 + conv2 + log_softmax (model/pointnet2.py:172-175) on position-major rows.

        l0_points [B,128,N] is a channel-first view of channel-last storage, so the rows view is free;
        conv1/bn1/relu and conv2 (128 -> classes) run through the same HIP GEMM kernels as the SA/FP stacks
        (the vendor GEMM needs 0.11 ms for the 13-column product), and the result is already in the reference's [B, N, classes] output layout.
        """
        B, C, N = l0_points.shape
        rows = l0_points.permute(0, 2, 1).reshape(B * N, C)
        feat = shared_mlp(rows, C, [self.conv1], [self.bn1], 0, self.training)
        x = conv1x1(self.drop1(feat), self.conv2, padded=True)             # [B*N, round4(classes)], pad columns zero
        return log_softmax_rows(x, self.conv2.out_channels).view(B, N, -1), feat.view(B, N, -1).permute(0, 2, 1)


class PointNet2ClsMsg(_ClsHead):
    """model/pointnet2.py:7-47."""

    def __init__(self):
        super().__init__()
        self.sa1 = PointNetSetAbstractionMsg(512, [0.1, 0.2, 0.4], [16, 32, 128], 0,
                                             [[32, 32, 64], [64, 64, 128], [64, 96, 128]])
        self.sa2 = PointNetSetAbstractionMsg(128, [0.2, 0.4, 0.8], [32, 64, 128], 320,
                                             [[64, 64, 128], [128, 128, 256], [128, 128, 256]])
        self.sa3 = PointNetSetAbstraction(None, None, None, 640 + 3, [256, 512, 1024], True)
        self._make_cls_head()

    def forward(self, xyz):
        l1_xyz, l1_points = self.sa1(xyz, None)
        l2_xyz, l2_points = self.sa2(l1_xyz, l1_points)
        _, l3_points = self.sa3(l2_xyz, l2_points)
        return self._cls_head(l3_points), l3_points


class PointNet2ClsSsg(_ClsHead):
    """model/pointnet2.py:49-73."""

    def __init__(self):
        super().__init__()
        self.sa1 = PointNetSetAbstraction(npoint=512, radius=0.2, nsample=32, in_channel=3, mlp=[64, 64, 128], group_all=False)
        self.sa2 = PointNetSetAbstraction(npoint=128, radius=0.4, nsample=64, in_channel=128 + 3, mlp=[128, 128, 256], group_all=False)
        self.sa3 = PointNetSetAbstraction(npoint=None, radius=None, nsample=None, in_channel=256 + 3, mlp=[256, 512, 1024], group_all=True)
        self._make_cls_head()

    def forward(self, xyz):
        l1_xyz, l1_points = self.sa1(xyz, None)
        l2_xyz, l2_points = self.sa2(l1_xyz, l1_points)
        _, l3_points = self.sa3(l2_xyz, l2_points)
        return self._cls_head(l3_points)


class PointNet2PartSegSsg(_SegHead):
    """model/pointnet2.py:75-104."""

    def __init__(self, num_classes):
        super().__init__()
        self.sa1 = PointNetSetAbstraction(npoint=512, radius=0.2, nsample=64, in_channel=3, mlp=[64, 64, 128], group_all=False)
        self.sa2 = PointNetSetAbstraction(npoint=128, radius=0.4, nsample=64, in_channel=128 + 3, mlp=[128, 128, 256], group_all=False)
        self.sa3 = PointNetSetAbstraction(npoint=None, radius=None, nsample=None, in_channel=256 + 3, mlp=[256, 512, 1024], group_all=True)
        self.fp3 = PointNetFeaturePropagation(in_channel=1280, mlp=[256, 256])
        self.fp2 = PointNetFeaturePropagation(in_channel=384, mlp=[256, 128])
        self.fp1 = PointNetFeaturePropagation(in_channel=128, mlp=[128, 128, 128])
        self._make_seg_head(num_classes)

    def forward(self, xyz):
        l1_xyz, l1_points = self.sa1(xyz, None)
        l2_xyz, l2_points = self.sa2(l1_xyz, l1_points)
        l3_xyz, l3_points = self.sa3(l2_xyz, l2_points)
        l2_points = self.fp3(l2_xyz, l3_xyz, l2_points, l3_points)
        l1_points = self.fp2(l1_xyz, l2_xyz, l1_points, l2_points)
        l0_points = self.fp1(xyz, l1_xyz, None, l1_points)
        return self._seg_head(l0_points)


class PointNet2PartSegMsg_one_hot(_SegHead):
    """model/pointnet2.py:106-139."""

    def __init__(self, num_classes):
        super().__init__()
        self.sa1 = PointNetSetAbstractionMsg(512, [0.1, 0.2, 0.4], [32, 64, 128], 0 + 3, [[32, 32, 64], [64, 64, 128], [64, 96, 128]])
        self.sa2 = PointNetSetAbstractionMsg(128, [0.4, 0.8], [64, 128], 128 + 128 + 64, [[128, 128, 256], [128, 196, 256]])
        self.sa3 = PointNetSetAbstraction(npoint=None, radius=None, nsample=None, in_channel=512 + 3, mlp=[256, 512, 1024], group_all=True)
        self.fp3 = PointNetFeaturePropagation(in_channel=1536, mlp=[256, 256])
        self.fp2 = PointNetFeaturePropagation(in_channel=576, mlp=[256, 128])
        self.fp1 = PointNetFeaturePropagation(in_channel=150, mlp=[128, 128])
        self._make_seg_head(num_classes)

    def forward(self, xyz, norm_plt, cls_label):
        B, _, N = xyz.size()
        l1_xyz, l1_points = self.sa1(xyz, norm_plt)
        l2_xyz, l2_points = self.sa2(l1_xyz, l1_points)
        l3_xyz, l3_points = self.sa3(l2_xyz, l2_points)
        l2_points = self.fp3(l2_xyz, l3_xyz, l2_points, l3_points)
        l1_points = self.fp2(l1_xyz, l2_xyz, l1_points, l2_points)
        one_hot = cls_label.view(B, 16, 1).repeat(1, 1, N)
        l0_points = self.fp1(xyz, l1_xyz, torch.cat([one_hot, xyz, norm_plt], 1), l1_points)
        return self._seg_head(l0_points)[0]


class PointNet2SemSeg(_SegHead):
    """model/pointnet2.py:141-176 -- the single-scale SemSeg network (SSG-SemSeg of the benchmark)."""

    def __init__(self, num_classes, feature_dims=3):
        super().__init__()
        self.feature_dims = feature_dims
        self.sa1 = PointNetSetAbstraction(1024, 0.1, 32, feature_dims + 3, [32, 32, 64], False)
        self.sa2 = PointNetSetAbstraction(256, 0.2, 32, 64 + 3, [64, 64, 128], False)
        self.sa3 = PointNetSetAbstraction(64, 0.4, 32, 128 + 3, [128, 128, 256], False)
        self.sa4 = PointNetSetAbstraction(16, 0.8, 32, 256 + 3, [256, 256, 512], False)
        self.fp4 = PointNetFeaturePropagation(768, [256, 256])
        self.fp3 = PointNetFeaturePropagation(384, [256, 256])
        self.fp2 = PointNetFeaturePropagation(320, [256, 128])
        self.fp1 = PointNetFeaturePropagation(128, [128, 128, 128])
        self._make_seg_head(num_classes)

    def features(self, points):
        """Everything on the hot path: the SA/FP stack up to l0_feature [B,128,N]."""
        xyz, feature = points[:, :3, :], points[:, 3:, :]
        l1_xyz, l1_feature = self.sa1(xyz, feature)
        l2_xyz, l2_feature = self.sa2(l1_xyz, l1_feature)
        l3_xyz, l3_feature = self.sa3(l2_xyz, l2_feature)
        l4_xyz, l4_feature = self.sa4(l3_xyz, l3_feature)
        l3_feature = self.fp4(l3_xyz, l4_xyz, l3_feature, l4_feature)
        l2_feature = self.fp3(l2_xyz, l3_xyz, l2_feature, l3_feature)
        l1_feature = self.fp2(l1_xyz, l2_xyz, l1_feature, l2_feature)
        return self.fp1(xyz, l1_xyz, None, l1_feature)

    def forward(self, points):
        return self._seg_head(self.features(points))[0]


class PointNet2SemSegMsg(_SegHead):
    """MSG-SemSeg: the MSG part-seg topology of model/pointnet2.py:106-139
    on [B, 3+D, N] inputs, without the one-hot label; 1 735 001 parameters at D = 6, 13 classes."""

    def __init__(self, num_classes, feature_dims=6, npoint_scale=1):
        super().__init__()
        d = feature_dims
        self.feature_dims = d
        self.sa1 = PointNetSetAbstractionMsg(512 * npoint_scale, [0.1, 0.2, 0.4], [32, 64, 128], d,
                                             [[32, 32, 64], [64, 64, 128], [64, 96, 128]])
        self.sa2 = PointNetSetAbstractionMsg(128 * npoint_scale, [0.4, 0.8], [64, 128], 128 + 128 + 64,
                                             [[128, 128, 256], [128, 196, 256]])
        self.sa3 = PointNetSetAbstraction(None, None, None, 512 + 3, [256, 512, 1024], True)
        self.fp3 = PointNetFeaturePropagation(1536, [256, 256])
        self.fp2 = PointNetFeaturePropagation(576, [256, 128])
        self.fp1 = PointNetFeaturePropagation(128 + 3 + d, [128, 128])
        self._make_seg_head(num_classes)

    def features(self, points):
        xyz, feature = points[:, :3, :], points[:, 3:, :]
        l1_xyz, l1_points = self.sa1(xyz, feature)
        l2_xyz, l2_points = self.sa2(l1_xyz, l1_points)
        l3_xyz, l3_points = self.sa3(l2_xyz, l2_points)
        l2_points = self.fp3(l2_xyz, l3_xyz, l2_points, l3_points)
        l1_points = self.fp2(l1_xyz, l2_xyz, l1_points, l2_points)
        return self.fp1(xyz, l1_xyz, points, l1_points)          # points == cat([xyz, feature], 1)

    def forward(self, points):
        return self._seg_head(self.features(points))[0]


def load_reference_state(model, state_dict):
    """Load a reference checkpoint (keys possibly prefixed ``module.`` by nn.DataParallel, model/utils.py:22)."""
    clean = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state_dict.items()}
    return model.load_state_dict(clean)
