"""PointCNN building blocks -- host-side mirror of /root/reference/misc/layers.py:112-517
(``EndChannels*`` :112-148, ``SepConv`` :133-169, ``Conv`` :173-205, ``Dense_Conv1d/2d`` :209-270,
``RandPointCNN_Decoder`` :273-303, ``RandPointCNN`` :306-336, ``PointCNN`` :340-411, ``XConv`` :415-517).

Layout.  The reference permutes every tensor to NCHW ``[B,C,P,K]`` so that Jittor's ``nn.Conv`` can be used, and
back again (``EndChannels``).  Here everything stays channel-last -- points ``[B,P,3]``, regions ``[B,P,K,C]`` -- so
the ``EndChannels`` wrappers vanish and each conv becomes a row-wise map:

  * ``Dense_Conv1d/2d`` (1x1 conv, bias, then BN, then ReLU, then dropout) = ``PointwiseMLP([cin,cout], bias=True)``
    on the fused gfx950 MLP kernels;
  * ``Conv(dims -> K*K, kernel (1,K))`` consumes a whole region per output pixel = one Linear over the flattened
    ``[K*dims]`` row of that region; reference weight ``w[o,d,0,k]`` is ``W[o, k*dims+d]`` here;
  * ``SepConv`` = depthwise ``(1,K)`` conv (channel c -> channels ``c*dm .. c*dm+dm-1``, reference weight
    ``wd[c*dm+j,0,0,k]`` is ``depthwise[c,j,k]`` here) followed by a 1x1 conv.

Order of activation and BatchNorm: ``Dense_*`` normalise THEN activate (:231-238); ``Conv`` and ``SepConv`` activate THEN
normalise, with ``momentum=0.9`` (:158-168, :196-204) -- kept as written.

The neighbourhoods come from the HIP KNN kernel (K*D nearest, every D-th kept, :396-400), the region gather from
``pcl_group_f32`` (which also emits the local coordinates ``pts - rep_pt`` of :474), sampling from the HIP FPS.
"""
import math

import torch
from torch import nn

from .. import _lib
from .layers import PointwiseMLP, batch_norm_train
from .ops import FurthestPointSampler, KNN, group_points, index_points, _p, _stream

__all__ = ["Dense_Conv1d", "Dense_Conv2d", "Conv", "SepConv", "XConv", "PointCNN", "RandPointCNN",
           "RandPointCNN_Decoder"]


class _Module(nn.Module):
    def execute(self, *a, **k):
        return self(*a, **k)


class _BatchNormLast(nn.Module):
    """BatchNorm over all leading dims of a channel-last tensor with Jittor's running-stat rule."""

    def __init__(self, channels, momentum=0.1, eps=1e-5):
        super().__init__()
        self.momentum, self.eps = momentum, eps
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))
        self.register_buffer("running_mean", torch.zeros(channels))
        self.register_buffer("running_var", torch.ones(channels))

    def forward(self, x):
        y = batch_norm_train(x.reshape(-1, x.shape[-1]), self.weight, self.bias, self.running_mean, self.running_var,
                             self.training, self.momentum, self.eps)
        return y.reshape(x.shape)


class Dense_Conv1d(_Module):
    """misc/layers.py:209-238.  ``[..., in_features] -> [..., out_features]``: Linear(bias) -> BN -> ReLU -> Dropout.
    ``activation``: ``"relu"`` or ``None`` (the reference passes an ``nn.ReLU()`` instance or ``None``)."""

    def __init__(self, in_features, out_features, drop_rate=0, with_bn=True, activation="relu"):
        super().__init__()
        self.mlp = PointwiseMLP([in_features, out_features], bias=True, bn=with_bn, last_act=activation is not None)
        self.drop = nn.Dropout(drop_rate) if drop_rate > 0 else None

    def forward(self, x):
        x = self.mlp(x.contiguous())
        return self.drop(x) if self.drop is not None else x


class Dense_Conv2d(Dense_Conv1d):
    """misc/layers.py:241-270 -- identical arithmetic on ``[B,P,K,C]`` (``groups`` is always 1 at the call sites)."""

    def __init__(self, in_features, out_features, drop_rate=0, with_bn=True, activation="relu", groups=1):
        if groups != 1:
            raise NotImplementedError("grouped Dense_Conv2d is not used by any network of the reference")
        super().__init__(in_features, out_features, drop_rate, with_bn, activation)


class Conv(_Module):
    """misc/layers.py:173-205 with ``kernel_size=(1,K)``: ``[B,P,K,in] -> [B,P,out]``; Linear over the flattened region
    (bias only without BN), ReLU, then BatchNorm(momentum 0.9)."""

    def __init__(self, in_channels, out_channels, kernel_size, with_bn=True, activation="relu"):
        super().__init__()
        kh, K = kernel_size
        assert kh == 1
        self.K, self.in_channels = K, in_channels
        self.linear = PointwiseMLP([K * in_channels, out_channels], bias=not with_bn, bn=False,
                                   last_act=activation is not None)
        self.bn = _BatchNormLast(out_channels, momentum=0.9) if with_bn else None

    def forward(self, x):
        B, P, K, C = x.shape
        assert K == self.K and C == self.in_channels
        y = self.linear(x.reshape(B, P, K * C))
        return self.bn(y) if self.bn is not None else y


class SepConv(_Module):
    """misc/layers.py:133-169 with ``kernel_size=(1,K)``: ``[B,P,K,C] -> [B,P,out]``; depthwise conv over the K axis
    (bias), 1x1 conv (bias only without BN), ReLU, BatchNorm(momentum 0.9)."""

    def __init__(self, in_channels, out_channels, kernel_size, depth_multiplier=1, with_bn=True, activation="relu"):
        super().__init__()
        kh, K = kernel_size
        assert kh == 1
        self.K, self.dm = K, depth_multiplier
        self.backend = "auto"                             # "auto" = the HIP kernels; any other name: a registered test reference
        bound = 1.0 / math.sqrt(K)                        # fan-in of a depthwise (1,K) filter
        self.depthwise = nn.Parameter(torch.empty(in_channels, depth_multiplier, K).uniform_(-bound, bound))
        self.depthwise_bias = nn.Parameter(torch.empty(in_channels * depth_multiplier).uniform_(-bound, bound))
        self.pointwise = PointwiseMLP([in_channels * depth_multiplier, out_channels], bias=not with_bn, bn=False,
                                      last_act=activation is not None)
        self.bn = _BatchNormLast(out_channels, momentum=0.9) if with_bn else None

    def forward(self, x):
        """``SepConv.execute`` (:160-169) on a region tensor [B,P,K,C]: the X-conv core below with X = I."""
        B, P, K, C = x.shape
        X = torch.eye(K, device=x.device, dtype=x.dtype).expand(B, P, K, K).contiguous()
        return self.forward_x(X, x, None)

    def forward_x(self, X, F1, F2):
        """``SepConv(X @ concat(F1, F2))`` with the matmul, the concat and the depthwise conv in one HIP kernel
        (csrc/xconv.hip); the pointwise conv runs on the MFMA GEMM as before."""
        if self.backend != "auto":       # a reference composite registered by test infrastructure (oracle/torch_backend.py)
            from .layers import reference_backend
            return reference_backend(self.backend).sepconv_forward_x(self, X, F1, F2)
        y = self.pointwise(xconv_core(X, F1, F2, self.depthwise, self.depthwise_bias))
        return self.bn(y) if self.bn is not None else y


class _XConvCore(torch.autograd.Function):
    """D[b,p,c*dm+j] = bias + sum_k wd[c,j,k] * (X[b,p] @ [F1|F2][b,p])[k,c]  (misc/layers.py:505 + the depthwise conv of :151)."""

    @staticmethod
    def forward(ctx, X, F1, F2, wd, bias):
        B, P, K, _ = X.shape
        C1, C2 = F1.shape[-1], (F2.shape[-1] if F2 is not None else 0)
        C, dm = wd.shape[0], wd.shape[1]
        assert C == C1 + C2 and wd.shape[2] == K
        X, F1, wd, bias = X.contiguous(), F1.contiguous(), wd.contiguous(), bias.contiguous()
        F2 = F2.contiguous() if F2 is not None else None
        D = torch.empty((B, P, C * dm), dtype=torch.float32, device=X.device)
        _lib.call("pcl_xconv_core_fwd_f32", _p(X), _p(F1), C1, _p(F2), C2, _p(wd), _p(bias), B * P, K, dm, _p(D), _stream(),
                  algo_bytes=4 * B * P * (K * C + K * K + C * dm), algo_flops=2 * B * P * C * K * (K + dm), tag=f"xconv{K}x{C}x{dm}")
        ctx.save_for_backward(X, F1, F2, wd)
        return D

    @staticmethod
    def backward(ctx, dD):
        X, F1, F2, wd = ctx.saved_tensors
        B, P, K, _ = X.shape
        C1, C2 = F1.shape[-1], (F2.shape[-1] if F2 is not None else 0)
        C, dm = wd.shape[0], wd.shape[1]
        dD = dD.contiguous()
        dev = X.device
        dX, dF1 = torch.empty_like(X), torch.empty_like(F1)
        dF2 = torch.empty_like(F2) if F2 is not None else None
        parts = _lib.size_query("pcl_xconv_core_partials", B * P, C)
        dwd_part = torch.empty((parts, C, dm, K), dtype=torch.float32, device=dev)
        db_part = torch.empty((parts, C * dm), dtype=torch.float32, device=dev)
        _lib.call("pcl_xconv_core_bwd_f32", _p(X), _p(F1), C1, _p(F2), C2, _p(wd), _p(dD), B * P, K, dm, _p(dX), _p(dF1), _p(dF2),
                  _p(dwd_part), _p(db_part), _stream(),
                  algo_bytes=4 * B * P * (2 * K * C + 2 * K * K + C * dm), algo_flops=2 * B * P * C * K * (3 * K + 2 * dm),
                  tag=f"xconvbwd{K}x{C}x{dm}")
        return dX, dF1, dF2, dwd_part.sum(0), db_part.sum(0)


def xconv_core(X, F1, F2, wd, bias):
    """[B,P,K,K], [B,P,K,C1], [B,P,K,C2]|None, taps [C1+C2, dm, K], bias [(C1+C2)*dm] -> [B,P,(C1+C2)*dm]."""
    if not X.is_cuda:
        raise RuntimeError("xconv_core needs GPU tensors (no CPU fallback)")
    K, C, dm = X.shape[-1], wd.shape[0], wd.shape[1]
    if not _lib.size_query("pcl_xconv_core_supported", K, dm, C):
        # shapes outside the kernel's instantiations (none in the reference's networks): the same arithmetic as separate GPU ops
        F = F1 if F2 is None else torch.cat((F1, F2), dim=-1)
        B, P = X.shape[:2]
        return torch.einsum("bpkc,cjk->bpcj", torch.matmul(X, F), wd).reshape(B, P, C * dm) + bias
    return _XConvCore.apply(X, F1, F2, wd, bias)


class XConv(_Module):
    """misc/layers.py:415-517.  ``(rep_pt[B,P,3], pts[B,P,K,3], fts[B,P,K,C_in]|None) -> [B,P,C_out]``."""

    def __init__(self, C_in, C_out, dims, K, P, C_mid, depth_multiplier):
        super().__init__()
        self.C_in, self.C_mid, self.dims, self.K, self.P = C_in, C_mid, dims, K, P
        self.dense = PointwiseMLP([dims, C_mid, C_mid], bias=True)        # dense1, dense2  :433-434
        self.x_trans_0 = Conv(dims, K * K, (1, K), with_bn=True)          # :437-441
        self.x_trans_1 = Dense_Conv2d(K * K, K * K, with_bn=True)         # :442
        self.x_trans_2 = Dense_Conv2d(K * K, K * K, with_bn=False, activation=None)   # :443
        self.end_conv = SepConv(C_mid + C_in, C_out, (1, K), depth_multiplier=depth_multiplier)   # :445-450

    def forward(self, x):
        rep_pt, pts, fts = x
        return self.forward_local(pts - rep_pt.unsqueeze(2), fts)          # :472-474

    def forward_local(self, pts_local, fts):
        B, P, K, dims = pts_local.shape
        assert K == self.K and dims == self.dims
        if fts is not None:
            assert fts.shape[:3] == (B, P, K) and fts.shape[3] == self.C_in
        pts_local = pts_local.contiguous()
        fts_lifted = self.dense(pts_local)                                 # [B,P,K,C_mid]  :480-483
        X = self.x_trans_2(self.x_trans_1(self.x_trans_0(pts_local)))     # [B,P,K*K]  :494-496
        X = X.reshape(B, P, K, K)                                          # :499-500
        # concat (:486-489), X @ fts_cat (:505) and the depthwise conv of end_conv (:509) in one kernel
        return self.end_conv.forward_x(X, fts_lifted, fts)


class PointCNN(_Module):
    """misc/layers.py:340-411.  ``(rep_pts[B,P,3], pts[B,N,3], fts[B,N,C_in]|None) -> [B,P,C_out]``."""

    def __init__(self, C_in, C_out, dims, K, D, P):
        super().__init__()
        C_mid = C_out // 2 if C_in == 0 else C_out // 4
        depth_multiplier = 4 if C_in == 0 else int(math.ceil(C_out / C_in))
        self.knn = KNN(K * D)
        self.dense = Dense_Conv1d(C_in, C_out // 2) if C_in != 0 else None
        self.x_conv = XConv(C_out // 2 if C_in != 0 else C_in, C_out, dims, K, P, C_mid, depth_multiplier)
        self.D, self.K = D, K

    def region_indices(self, rep_pts, pts):
        """int32 [B,P,K]: the K*D nearest points of each representative, every D-th kept (:396-400)."""
        idx = self.knn(rep_pts.transpose(1, 2).contiguous(), pts.transpose(1, 2).contiguous())   # [B,K*D,P]
        return idx[:, 0::self.D, :].permute(0, 2, 1).contiguous()

    def forward(self, x):
        rep_pts, pts, fts = x
        fts = self.dense(fts) if fts is not None else None                 # :393
        idx = self.region_indices(rep_pts, pts)
        pts_local = group_points(pts.contiguous(), rep_pts.contiguous(), None, idx, use_xyz=True)   # [B,P,K,3] :401,:474
        fts_regional = index_points(fts, idx) if fts is not None else None                         # [B,P,K,C] :402
        return self.x_conv.forward_local(pts_local, fts_regional)


class RandPointCNN(_Module):
    """misc/layers.py:306-336: FPS-subsampled representatives (``0 < P < N``), else every point."""

    def __init__(self, C_in, C_out, dims, K, D, P):
        super().__init__()
        self.pointcnn = PointCNN(C_in, C_out, dims, K, D, P)
        self.P = P
        if self.P > 0:
            self.sampler = FurthestPointSampler(self.P)

    def forward(self, x):
        pts, fts = x
        rep_pts = self.sampler(pts) if 0 < self.P < pts.shape[1] else pts
        return rep_pts, self.pointcnn((rep_pts, pts, fts))


class RandPointCNN_Decoder(_Module):
    """misc/layers.py:273-303: X-conv from the coarse level onto the fine level's points, concat, fuse."""

    def __init__(self, C_in, C_out, C_last, dims, K, D, P):
        super().__init__()
        self.pointcnn = PointCNN(C_in, C_out, dims, K, D, P)
        self.P = P
        self.conv_fuse = Dense_Conv1d(C_out + C_last, C_out)

    def forward(self, x_l, x_h):
        pts_l, fts_l = x_l
        pts_h, fts_h = x_h
        rep_pts_fts = self.pointcnn((pts_h, pts_l, fts_l))
        return pts_h, self.conv_fuse(torch.cat((rep_pts_fts, fts_h), dim=2))
