"""DGCNN part segmentation -- counterpart of /root/reference/networks/seg/dgcnn_partseg.py:36-128.

Three EdgeConv stages with k=40 (two 1x1 convs in the first two, one in the third, each + BN + LeakyReLU(0.2), max
over k), conv6 192->1024 + global max, the 16-way object label lifted to 64 channels by conv7, broadcast and
concatenated with x1..x3 (1088+192 = 1280), conv8-10 with dropout, conv11 to the part logits.  Input ``x [B,3,N]``,
``l [B,16]``; output ``[B,part_num,N]``.  kNN, the edge-feature gather and every conv stack run on the HIP kernels."""
import torch
from torch import nn

from ...misc.layers import PointwiseMLP
from ...misc.ops import KNN
from ...misc.edgeconv import edge_conv
from ..cls.dgcnn import get_graph_feature, knn_graph


class DGCNN_partseg(nn.Module):
    def __init__(self, part_num):
        super().__init__()
        self.seg_num_all = part_num
        self.k = 40
        self.knn = KNN(self.k)
        self.conv12 = PointwiseMLP([6, 64, 64], slope=0.2)            # conv1, conv2   :54-59
        self.conv34 = PointwiseMLP([64 * 2, 64, 64], slope=0.2)       # conv3, conv4   :60-65
        self.conv5 = PointwiseMLP([64 * 2, 64], slope=0.2)            # :66-68
        self.conv6 = PointwiseMLP([192, 1024], slope=0.2)             # :69-71
        self.conv7 = PointwiseMLP([16, 64], slope=0.2)                # :72-74
        self.conv8 = PointwiseMLP([1280, 256], slope=0.2)             # :75-77
        self.dp1 = nn.Dropout(p=0.5)
        self.conv9 = PointwiseMLP([256, 256], slope=0.2)              # :79-81
        self.dp2 = nn.Dropout(p=0.5)
        self.conv10 = PointwiseMLP([256, 128], slope=0.2)             # :83-85
        self.conv11 = nn.Linear(128, part_num, bias=False)            # :86

    def forward(self, x, l, lists=None, return_stages=False):
        """``lists``: optional neighbour lists [3] x int32 [B,N,k] used instead of the network's own kNN -- the reference's
        ``get_graph_feature(..., idx=...)`` argument (:11,:15): whole-network comparisons between two fp32 pipelines share one
        set of lists (a near-tie in a FEATURE-space kNN orders two neighbours differently for inputs that agree to 1e-6)."""
        B, _, N = x.shape
        x = x.transpose(1, 2).contiguous()                                         # channel-last [B,N,3]
        g = (lambda i, t: knn_graph(t, self.knn)) if lists is None else (lambda i, t: lists[i])
        x1 = self.conv12(get_graph_feature(x, idx=g(0, x)), group_max=self.k)      # :95-98
        x2 = self.conv34(get_graph_feature(x1, idx=g(1, x1)), group_max=self.k)    # :100-103
        x3 = edge_conv(self.conv5, x2, g(2, x2))                                   # :105-107 (single conv: factorised)
        x123 = torch.cat((x1, x2, x3), dim=2)                                      # [B,N,192]  :109
        g = self.conv6(x123[:, None].contiguous(), group_max=N).reshape(B, 1024)   # conv6 + max over N  :111-112
        lf = self.conv7(l.reshape(B, 16))                                          # :114-115
        glob = torch.cat((g, lf), dim=1)[:, None, :].expand(B, N, 1088)            # :117-118
        y = torch.cat((glob, x123), dim=2).contiguous()                            # :120
        y = self.dp1(self.conv8(y))
        y = self.dp2(self.conv9(y))
        y = self.conv10(y)
        out = self.conv11(y).permute(0, 2, 1)                                      # [B,part,N]
        return (out, (x1, x2, x3)) if return_stages else out

    def execute(self, *a, **k):
        return self(*a, **k)
