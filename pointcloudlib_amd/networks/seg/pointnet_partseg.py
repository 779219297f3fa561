"""PointNet part segmentation -- counterpart of /root/reference/networks/seg/pointnet_partseg.py:14-66.

Input transform (STN3d), conv 3->64->128->128, feature transform (STNkd 128), conv 128->512->2048 (the last one BN
only, :54), global max, concat with the 16-way object label, broadcast to every point and concatenated with all five
per-point activations (2064+64+128+128+512+2048 = 4944, :60-61), head 4944->256->256->128->part_num.
``point_cloud`` is ``[B,3,N]`` and the output ``[B,part_num,N]`` like the reference; internally channel-last."""
import torch
from torch import nn

from ...misc.layers import PointwiseMLP
from ...misc.stn import STN3d, STNkd


class PointNet_partseg(nn.Module):
    def __init__(self, part_num=50):
        super().__init__()
        self.part_num = part_num
        self.stn = STN3d()
        self.conv1 = PointwiseMLP([3, 64], bias=True)
        self.conv2 = PointwiseMLP([64, 128], bias=True)
        self.conv3 = PointwiseMLP([128, 128], bias=True)
        self.fstn = STNkd(k=128)
        self.conv4 = PointwiseMLP([128, 512], bias=True)
        self.conv5 = PointwiseMLP([512, 2048], bias=True, last_act=False)     # bn5(conv5(.)) without ReLU  :54
        self.convs = PointwiseMLP([4944, 256, 256, 128], bias=True)           # convs1-3 + bns1-3 + relu    :62-64
        self.convs4 = nn.Linear(128, part_num)

    def forward(self, point_cloud, label):
        B, D, N = point_cloud.shape
        pc = point_cloud.transpose(1, 2).contiguous()                          # [B,N,3]
        pc = torch.bmm(pc, self.stn(pc))                                       # :43-45
        out1 = self.conv1(pc)
        out2 = self.conv2(out1)
        out3 = self.conv3(out2)
        net_transformed = torch.bmm(out3, self.fstn(out3))                     # :51-53
        out4 = self.conv4(net_transformed)
        out5 = self.conv5(out4)
        out_max = out5.max(dim=1)[0]                                           # :56-57
        expand = torch.cat((out_max, label), 1)[:, None, :].expand(B, N, 2048 + 16)   # :59-60
        net = self.convs(torch.cat([expand, out1, out2, out3, out4, out5], 2).contiguous())
        return self.convs4(net).permute(0, 2, 1)

    def execute(self, *a, **k):
        return self(*a, **k)
