"""PointNet++ part segmentation -- counterpart of /root/reference/networks/seg/pointnet2_partseg.py
(PointNet2_partseg :110-176, PointNetMSG :179-214).

Encoder = the classification SA modules, except that the GroupAll stage reports new_xyz = zeros[B,1,3] (:55);
decoder = three PointNetFeaturePropagation (:146-148, misc/ops.py:54-107); head Conv1d 128->128 + BN1d (no ReLU)
+ Dropout 0.5 + Conv1d 128->part_num (:151-156).  Output [B,part_num,N] like the reference's Conv1d head.

Deviation from an upstream bug: upstream's PointNetMSG keeps the SSG FP widths 1280/384,
which cannot match its MSG encoder (640+1024, 320+256); here the MSG variant uses the consistent 1664/576.
"""
import os

import torch
from torch import nn

from ...misc.layers import PointwiseMLP, set_accumulation
from ...misc.ops import PointNetFeaturePropagation
from ..cls.pointnet2 import PointnetModule, PointnetModuleMSG, SamplingPrefetch


class PointNet2_partseg(SamplingPrefetch, nn.Module):
    def __init__(self, part_num=50, use_xyz=True):
        super().__init__()
        self.part_num = part_num
        self.use_xyz = use_xyz
        self.build_model()
        # The decoder stacks up to 14 BatchNorms on the encoder's output; with one fp32 fma chain per dot product (K up to 1664) the
        # logits sit 6-8 x the 1e-5 bound from the fp64 evaluation, 4-5 x of it accumulation error (tools/dbg/partseg_local_err.py).  Every
        # forward GEMM on plain rows (GroupAll level, feature propagation, head, the per-point products of the grouped levels) therefore
        # sums chains of 32 terms in fp64 (csrc/frag.hip); PCL_PARTSEG_FLUSH=0 selects the plain kernels (A/B timing).
        set_accumulation(self, int(os.environ.get("PCL_PARTSEG_FLUSH", "32")))

    def build_model(self):
        self.pointnet_modules = nn.ModuleList([
            PointnetModule(n_points=512, radius=0.2, n_samples=64, mlp=[3, 64, 64, 128], use_xyz=self.use_xyz),
            PointnetModule(n_points=128, radius=0.4, n_samples=64, mlp=[128, 128, 128, 256], use_xyz=self.use_xyz),
            PointnetModule(mlp=[256, 256, 512, 1024], use_xyz=self.use_xyz),
        ])
        self.fp3 = PointNetFeaturePropagation(in_channel=1280, mlp=[256, 256])
        self.fp2 = PointNetFeaturePropagation(in_channel=384, mlp=[256, 128])
        self.fp1 = PointNetFeaturePropagation(in_channel=128 + 16 + 6, mlp=[128, 128, 128])
        self.build_head()

    def build_head(self):
        # Conv1d(128,128,1) + BatchNorm1d(128) [no activation] ; Dropout ; Conv1d(128, part_num, 1)   :151-156
        self.head1 = PointwiseMLP([128, 128], bias=True, slope=0.0, last_act=False)
        self.drop = nn.Dropout(0.5)
        self.head2 = PointwiseMLP([128, self.part_num], bias=True, bn=False, last_act=False)   # Conv1d(128, part_num, 1)

    def forward(self, xyz, feature, cls_label, sampling=None):
        """xyz [B,N,3], feature [B,N,3], cls_label one-hot [B,16] -> [B,part_num,N].  ``sampling``: a handle from
        ``precompute_sampling(xyz)`` (the encoder's FPS / ball-query indices produced ahead, e.g. on a side stream)."""
        B, N, _ = xyz.shape
        self.adopt_sampling(sampling)
        lv = [None, None, None] if sampling is None else sampling["levels"]
        l1_xyz, l1_feature = self.pointnet_modules[0](xyz, feature, lv[0])
        l2_xyz, l2_feature = self.pointnet_modules[1](l1_xyz, l1_feature, lv[1])
        _, l3_feature = self.pointnet_modules[2](l2_xyz, l2_feature, lv[2])
        l3_xyz = torch.zeros((B, 1, 3), device=xyz.device, dtype=xyz.dtype)          # :55
        l2_feature = self.fp3(l2_xyz, l3_xyz, l2_feature, l3_feature)                 # :168
        l1_feature = self.fp2(l1_xyz, l2_xyz, l1_feature, l2_feature)                 # :169
        one_hot = cls_label.view(B, 1, 16).expand(B, N, 16)                           # :170
        feature = self.fp1(xyz, l1_xyz, torch.cat([one_hot, xyz, feature], 2), l1_feature)   # :173
        x = self.drop(self.head1(feature))
        return self.head2(x).permute(0, 2, 1)

    def execute(self, *a, **k):
        return self(*a, **k)


class PointNetMSG(PointNet2_partseg):
    def build_model(self):
        self.pointnet_modules = nn.ModuleList([
            PointnetModuleMSG(n_points=512, radius=[0.1, 0.2, 0.4], n_samples=[16, 32, 128],
                              mlps=[[3, 32, 32, 64], [3, 64, 64, 128], [3, 64, 96, 128]], use_xyz=self.use_xyz),
            PointnetModuleMSG(n_points=128, radius=[0.2, 0.4, 0.8], n_samples=[32, 64, 128],
                              mlps=[[320, 64, 64, 128], [320, 128, 128, 256], [320, 128, 128, 256]], use_xyz=self.use_xyz),
            PointnetModule(mlp=[128 + 256 + 256, 256, 512, 1024], use_xyz=self.use_xyz),
        ])
        self.fp3 = PointNetFeaturePropagation(in_channel=640 + 1024, mlp=[256, 256])
        self.fp2 = PointNetFeaturePropagation(in_channel=320 + 256, mlp=[256, 128])
        self.fp1 = PointNetFeaturePropagation(in_channel=128 + 16 + 6, mlp=[128, 128, 128])
        self.build_head()
