"""PointConv part segmentation -- counterpart of /root/reference/networks/seg/pointconv_partseg.py:9-63.

Four ``PointConvDensitySetAbstraction`` levels (1024/256/64/36 points, 32 neighbours), four
``PointConvDensitySetInterpolation`` levels back up (16 neighbours), head Conv1d 128->128 + BN + ReLU + Dropout 0.4
+ Conv1d 128->part_num.  ``xyz`` is ``[B,N,3]`` (permuted at :42) and the output ``[B,N,part_num]`` (:61);
``cls_label`` is accepted and unused, as upstream."""
from torch import nn

from ...misc.layers import PointwiseMLP
from ...misc.pointconv_utils import PointConvDensitySetAbstraction, PointConvDensitySetInterpolation


class PointConvDensity_partseg(nn.Module):
    def __init__(self, part_num=50):
        super().__init__()
        self.part_num = part_num
        SA, IN = PointConvDensitySetAbstraction, PointConvDensitySetInterpolation
        self.sa0 = SA(npoint=1024, nsample=32, in_channel=3, mlp=[32, 32, 64], bandwidth=0.1, group_all=False)
        self.sa1 = SA(npoint=256, nsample=32, in_channel=64 + 3, mlp=[64, 64, 128], bandwidth=0.2, group_all=False)
        self.sa2 = SA(npoint=64, nsample=32, in_channel=128 + 3, mlp=[128, 128, 256], bandwidth=0.4, group_all=False)
        self.sa3 = SA(npoint=36, nsample=32, in_channel=256 + 3, mlp=[256, 256, 512], bandwidth=0.8, group_all=False)
        self.in0 = IN(nsample=16, in_channel=512 + 3, mlp=[512, 512], bandwidth=0.8)
        self.in1 = IN(nsample=16, in_channel=512 + 3, mlp=[256, 256], bandwidth=0.4)
        self.in2 = IN(nsample=16, in_channel=256 + 3, mlp=[128, 128], bandwidth=0.2)
        self.in3 = IN(nsample=16, in_channel=128 + 3, mlp=[128, 128, 128], bandwidth=0.1)
        self.fc1 = PointwiseMLP([128, 128], bias=True)               # fc1 + bn1 + relu  :36-37,:58
        self.drop1 = nn.Dropout(0.4)
        self.fc3 = nn.Linear(128, part_num)

    def forward(self, xyz, cls_label=None):
        xyz = xyz.permute(0, 2, 1).contiguous()                       # [B,3,N]  :42
        l1_xyz, l1_points = self.sa0(xyz, None)
        l2_xyz, l2_points = self.sa1(l1_xyz, l1_points)
        l3_xyz, l3_points = self.sa2(l2_xyz, l2_points)
        l4_xyz, l4_points = self.sa3(l3_xyz, l3_points)
        l3_points = self.in0(l3_xyz, l4_xyz, l3_points, l4_points)    # :51-54
        l2_points = self.in1(l2_xyz, l3_xyz, l2_points, l3_points)
        l1_points = self.in2(l1_xyz, l2_xyz, l1_points, l2_points)
        l0_points = self.in3(xyz, l1_xyz, xyz, l1_points)
        x = self.drop1(self.fc1(l0_points.permute(0, 2, 1).contiguous()))   # [B,N,128]
        return self.fc3(x)                                            # [B,N,part_num]  :59-61

    def execute(self, *a, **k):
        return self(*a, **k)
