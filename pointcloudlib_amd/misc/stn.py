"""PointNet's T-Nets -- host-side mirror of ``STN3d`` / ``STNkd`` (/root/reference/misc/layers.py:11-87).

conv 1x1 (bias) + BatchNorm1d + ReLU: k->64->128->1024, max over the points, fc 1024->512->256 (+BN+ReLU), fc3 to
k*k, plus the identity.  Input channel-last ``[B,N,k]`` (the reference takes ``[B,k,N]``); output ``[B,k,k]``
(``STNkd.execute`` upstream returns the flat ``[B,k*k]`` and lets ``nn.bmm`` fail on it -- reshaped here, as
``STN3d`` does at :56)."""
import torch
from torch import nn

from .layers import PointwiseMLP


class STNkd(nn.Module):
    def __init__(self, k=64):
        super().__init__()
        self.k = k
        self.convs = PointwiseMLP([k, 64, 128, 1024], bias=True)      # conv1-3 + bn1-3 + relu
        self.fcs = PointwiseMLP([1024, 512, 256], bias=True)          # fc1-2 + bn4-5 + relu
        self.fc3 = nn.Linear(256, k * k)

    def forward(self, x):
        B, N, k = x.shape
        assert k == self.k
        g = self.convs(x[:, None].contiguous(), group_max=N).reshape(B, 1024)   # jt.max(x, 2)  :39,:81
        t = self.fc3(self.fcs(g))
        iden = torch.eye(k, device=x.device, dtype=x.dtype).reshape(1, k * k)
        return (t + iden).reshape(B, k, k)

    def execute(self, *a, **k):
        return self(*a, **k)


class STN3d(STNkd):
    def __init__(self):
        super().__init__(k=3)
