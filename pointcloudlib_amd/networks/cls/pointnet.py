"""PointNet classification -- counterpart of /root/reference/networks/cls/pointnet.py:10-40 (BASELINE config 1:
no sampling/grouping; exercises the conv1d/BN/max/FC plumbing).  Input [B,3,N] like the reference."""
from torch import nn

from ...misc.head import fc_head
from ...misc.layers import PointwiseMLP


class PointNet(nn.Module):
    def __init__(self, output_channels=40):
        super().__init__()
        self.convs = PointwiseMLP([3, 64, 64, 64, 128, 1024], bias=False)      # conv1-5 + bn1-5 + relu  :12-21
        self.linear1 = nn.Linear(1024, 512, bias=False)
        self.bn6 = nn.BatchNorm1d(512)
        self.dp1 = nn.Dropout(0.5)
        self.linear2 = nn.Linear(512, output_channels)
        self.relu = nn.ReLU()

    def forward(self, x):
        x = x.transpose(1, 2).contiguous()                # [B,N,3]
        x = self.convs(x[:, None], group_max=x.shape[1])  # conv stack + max over N   :30-35  -> [B,1,1024]
        x = x.reshape(x.shape[0], -1)
        return fc_head([self.linear1, self.bn6, self.relu, self.dp1, self.linear2], x)     # one call per direction (misc/head.py)

    def execute(self, *a, **k):
        return self(*a, **k)
