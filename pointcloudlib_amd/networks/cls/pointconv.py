"""PointConv classification -- counterpart of /root/reference/networks/cls/pointconv.py:8-34 (BASELINE config 5).
sa3 uses ``group_all=True``, which upstream cannot run (missing ``sample_and_group_all``); see pointconv_utils."""
from torch import nn

from ...misc.head import fc_head
from ...misc.pointconv_utils import PointConvDensitySetAbstraction


class PointConvDensityClsSsg(nn.Module):
    def __init__(self, n_classes=40):
        super().__init__()
        self.sa1 = PointConvDensitySetAbstraction(npoint=512, nsample=32, in_channel=3, mlp=[64, 64, 128], bandwidth=0.1, group_all=False)
        self.sa2 = PointConvDensitySetAbstraction(npoint=128, nsample=64, in_channel=128 + 3, mlp=[128, 128, 256], bandwidth=0.2, group_all=False)
        self.sa3 = PointConvDensitySetAbstraction(npoint=1, nsample=None, in_channel=256 + 3, mlp=[256, 512, 1024], bandwidth=0.4, group_all=True)
        self.fc1 = nn.Linear(1024, 512)
        self.bn1 = nn.BatchNorm1d(512)
        self.drop1 = nn.Dropout(0.4)
        self.fc2 = nn.Linear(512, 256)
        self.bn2 = nn.BatchNorm1d(256)
        self.drop2 = nn.Dropout(0.4)
        self.fc3 = nn.Linear(256, n_classes)
        self.relu = nn.ReLU()

    def forward(self, xyz, start_idx=None, knn_lists=None):
        """xyz [B,3,N] (the reference permutes from [B,N,3] at :26; callers here pass [B,3,N] directly).
        ``knn_lists``: optional neighbour groups (int32 [B,512,32], [B,128,64]) replacing ``knn_point``'s."""
        B = xyz.shape[0]
        k1, k2 = (None, None) if knn_lists is None else knn_lists
        l1_xyz, l1_points = self.sa1(xyz, None, None if start_idx is None else start_idx[0], k1)
        l2_xyz, l2_points = self.sa2(l1_xyz, l1_points, None if start_idx is None else start_idx[1], k2)
        _, l3_points = self.sa3(l2_xyz, l2_points)
        x = l3_points.reshape(B, 1024)
        # one call per direction (misc/head.py: fc_head -> pcl_fc_head_*_f32)
        return fc_head([self.fc1, self.bn1, self.relu, self.drop1, self.fc2, self.bn2, self.relu, self.drop2, self.fc3], x)

    def execute(self, *a, **k):
        return self(*a, **k)
