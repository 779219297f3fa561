"""PointCNN classification -- counterpart of /root/reference/networks/cls/pointcnn.py:22-53.

``AbbPointCNN(a, b, c, d, e) = RandPointCNN(C_in=a, C_out=b, dims=3, K=c, D=d, P=e)`` (:20); four X-conv stages
3->48 (K=8, all points), 48->96 (K=12, D=2, 384 FPS points), 96->192 (K=16, D=2, 128), 192->384 (K=16, D=3, 128);
head ``Dense_Conv1d`` 384->192->128(dropout 0.5)->n_classes, logits = mean over the remaining points (:50-52).
Activations are channel-last ``[B,P,C]`` throughout (the reference permutes to ``[B,C,P]`` for its head, :49).
"""
from torch import nn

from ...misc.pointcnn import Dense_Conv1d, RandPointCNN


def AbbPointCNN(a, b, c, d, e):
    return RandPointCNN(a, b, 3, c, d, e)


class PointCNNcls(nn.Module):
    def __init__(self, n_classes=40):
        super().__init__()
        self.pcnn1 = AbbPointCNN(3, 48, 8, 1, -1)
        self.pcnn2 = nn.Sequential(
            AbbPointCNN(48, 96, 12, 2, 384),
            AbbPointCNN(96, 192, 16, 2, 128),
            AbbPointCNN(192, 384, 16, 3, 128),
        )
        self.fcn = nn.Sequential(
            Dense_Conv1d(384, 192),
            Dense_Conv1d(192, 128, drop_rate=0.5),
            Dense_Conv1d(128, n_classes, with_bn=False, activation=None),
        )

    def forward(self, x, normal=None):
        """x [B,N,3] (and optional per-point features ``normal`` [B,N,3]) -> logits [B,n_classes]."""
        x = (x, x if normal is None else normal)       # :40-43
        x = self.pcnn1(x)
        x = self.pcnn2(x)[1]                           # features [B,128,384]
        logits = self.fcn(x)                           # [B,128,n_classes]
        return logits.mean(dim=1)                      # :51

    def execute(self, *a, **k):
        return self(*a, **k)
