"""DGCNN classification -- counterpart of /root/reference/networks/cls/dgcnn.py (DGCNN :61-122,
get_graph_feature :29-50).

Same layer widths, k and activations: four EdgeConv stages (kNN in the CURRENT feature space -> edge features
concat(nbr - ctr, ctr) -> Conv2d 1x1 (bias=False) + BN + LeakyReLU(0.2) -> max over k) :72-83,:100-111;
concat(x1..x4) -> Conv1d 512->1024 + BN1d + LeakyReLU :84-86,:113; global max || global mean :114-116; FC head
:87-93,:117-121.  Input is [B,3,N] like the reference; internally activations are channel-last [B,N,C].

kNN runs in the HIP KNN kernel (index-exact vs the oracle).  The EdgeConv stages use the factorised form of
misc/edgeconv.py (one GEMM over the points + two streaming kernels, no edge tensor); ``get_graph_feature`` -- the
reference's edge tensor, via pcl_edge_feature_f32 -- stays available and is what the plain-PyTorch backend uses.
"""
import os

import torch
from torch import nn

from ...misc.layers import PointwiseMLP
from ...misc.edgeconv import assemble, conv_max_mean_pool, edge_conv
from ...misc.head import fc_head
from ...misc.ops import KNN, edge_features, knn_lists


def knn_graph(x, knn, xt=None):
    """x [B,N,C] channel-last -> neighbour lists int32 [B,N,k] in the CURRENT feature space (dgcnn.py:34-35).  ``xt``: x as [B,C,N] where
    the producing stage handed that layout on as well (``edge_conv(..., want_t=True)``)."""
    if xt is None:
        xt = x.transpose(1, 2).contiguous()           # [B,C,N] as KNN expects (misc/ops.py:651)
    if xt.is_cuda:
        return knn_lists(xt, xt, knn.k)               # the same lists, written as [B,N,k] rows by the search itself (no permute copy)
    return knn(xt, xt).permute(0, 2, 1).contiguous()


def get_graph_feature(x, knn=None, k=None, idx=None):
    """x [B,N,C] channel-last -> [B,N,k,2C]  (dgcnn.py:29-50; the reference returns [B,2C,N,k])."""
    if idx is None:
        xt = x.transpose(1, 2).contiguous()           # [B,C,N] as KNN expects (misc/ops.py:651)
        idx = knn(xt, xt).permute(0, 2, 1).contiguous()   # [B,k,N] -> [B,N,k]   dgcnn.py:34-35
    return edge_features(x, idx)


_STAGE_T = os.environ.get("PCL_DGCNN_STAGE_T", "1") != "0"                # lab switch: 0 = a transpose launch in front of every search
_CAT_IN_PLACE = os.environ.get("PCL_DGCNN_CAT_IN_PLACE", "1") != "0"      # lab switch (A/B on one box): 0 = torch.cat
_LRELU = nn.LeakyReLU(0.2)          # activation marker for fc_head (no parameters, not part of the state dict)


class DGCNN(nn.Module):
    def __init__(self, n_classes=40):
        super().__init__()
        self.k = 20
        self.knn = KNN(self.k)
        self.conv1 = PointwiseMLP([6, 64], slope=0.2)
        self.conv2 = PointwiseMLP([64 * 2, 64], slope=0.2)
        self.conv3 = PointwiseMLP([64 * 2, 128], slope=0.2)
        self.conv4 = PointwiseMLP([128 * 2, 256], slope=0.2)
        self.conv5 = PointwiseMLP([512, 1024], slope=0.2)
        # U | V of the EdgeConv stages decide the max-pool winners: chains of 8 terms summed in fp64 (misc/edgeconv.py: _PointLinear)
        # (PCL_DGCNN_HILO=1 also carries the products' residuals into the gather: exact on equal inputs, +5 % per step -- misc/edgeconv.py)
        for c in (self.conv1, self.conv2, self.conv3, self.conv4):
            c.flush_k = int(os.environ.get("PCL_DGCNN_FLUSH", "8"))
            c.edge_hilo = os.environ.get("PCL_DGCNN_HILO", "0") != "0"

        self.linear1 = nn.Linear(1024 * 2, 512, bias=False)
        self.bn6 = nn.BatchNorm1d(512)
        self.dp1 = nn.Dropout(p=0.5)
        self.linear2 = nn.Linear(512, 256)
        self.bn7 = nn.BatchNorm1d(256)
        self.dp2 = nn.Dropout(p=0.5)
        self.linear3 = nn.Linear(256, n_classes)

    def forward(self, x, lists=None, return_stages=False):
        """x [B,3,N] -> logits [B,n_classes].  ``lists``: optional neighbour lists [4] x int32 [B,N,k] used instead of the
        network's own kNN (the reference's ``get_graph_feature(..., idx=...)`` argument, dgcnn.py:29,33): whole-network
        comparisons between two fp32 pipelines share one set of lists, because a near-tie in a FEATURE-space kNN orders
        two neighbours differently for inputs that agree to 1e-6."""
        x_in = x                                                           # [B,3,N]: the layout the first search reads
        x = x.transpose(1, 2).contiguous()                                 # channel-last
        wt = lists is None and x.is_cuda and _STAGE_T                      # stages hand their output on in both layouts: no transpose launches
        g = (lambda i, t, tt=None: knn_graph(t, self.knn, tt)) if lists is None else (lambda i, t, tt=None: lists[i])
        # concat(x1..x4) [B,N,512] (:112) is written by the stages themselves (each output goes to its column slice as well): no copy kernel
        cat = torch.empty((x.shape[0], x.shape[1], 512), dtype=x.dtype, device=x.device) if (x.is_cuda and _CAT_IN_PLACE) else None
        sl = (lambda a, b: None) if cat is None else (lambda a, b: cat[:, :, a:b])
        pair = (lambda r: r) if wt else (lambda r: (r, None))
        x1, t1 = pair(edge_conv(self.conv1, x, g(0, x, x_in if wt else None), sl(0, 64), wt))      # :100-102
        x2, t2 = pair(edge_conv(self.conv2, x1, g(1, x1, t1), sl(64, 128), wt))                    # :103-105
        x3, t3 = pair(edge_conv(self.conv3, x2, g(2, x2, t2), sl(128, 256), wt))                   # :106-108
        x4 = edge_conv(self.conv4, x3, g(3, x3, t3), sl(256, 512))                                  # :109-111
        stages = (x1, x2, x3, x4)
        x = torch.cat((x1, x2, x3, x4), dim=2) if cat is None else assemble(cat, stages)   # [B,N,512]   :112
        x = conv_max_mean_pool(self.conv5, x)                               # conv5 + max / mean over the points + concat  :113-116
        # :117-121 -- the whole head as one call per direction (misc/head.py: fc_head -> pcl_fc_head_*_f32)
        x = fc_head([self.linear1, self.bn6, _LRELU, self.dp1, self.linear2, self.bn7, _LRELU, self.dp2, self.linear3], x)
        return (x, stages) if return_stages else x

    def execute(self, *a, **k):
        return self(*a, **k)
