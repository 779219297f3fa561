"""PointConv building blocks -- host-side mirror of /root/reference/misc/pointconv_utils.py.

Same function/class names and argument meaning: ``farthest_point_sample`` :74, ``knn_point`` :120,
``sample_and_group`` :133, ``compute_density`` :174, ``DensityNet`` :186, ``WeightNet`` :220,
``PointConvDensitySetInterpolation`` :252, ``PointConvDensitySetAbstraction`` :340.  All tensors channel-last ([B,N,C]); the reference's permutes to
[B,C,ns,npoint] around its Conv2d(k=1) stacks disappear (1x1 conv == row-wise linear map).

What runs where: FPS (no origin skip, caller-supplied start index -- the reference draws it with
np.random.randint, :88), k-NN grouping (HIP KNN kernel, direct-form distances, ties -> lower index; the reference
uses matmul-form distances + a full argsort, :120-131 -- parity unpinned, documented), gathers, Gaussian KDE
(pcl_density_f32, no [B,N,N] matrix), every Conv+BN+ReLU stack (fused MFMA MLP) and the density-weighted per-point
(C x ns)(ns x 16) contraction (pcl_pointconv_contract_f32, one streaming pass instead of multiply + transpose + bmm) are HIP.

Upstream bugs handled: ``sample_and_group_all`` is called at :380 but defined nowhere --
implemented here with the original PointConv semantics (one group of all N points, xyz relative to the cloud
centroid, density reshaped [B,1,N,1]); ``DensityNet``'s sigmoid branch (:213) is unreachable upstream (``i == len``)
so every layer is BN+ReLU, reproduced as such.
"""
import os

import torch
from torch import nn

from .. import _lib
from .layers import PointwiseMLP
from .ops import _dev, _p, _stream, furthest_point_sample, index_points, knn_indices, three_interpolate, three_nn

# Accumulation of the per-point Linear(16 C -> C) of a PointConv level (misc/pointconv_utils.py:395-397).  Its dot products are 16 C =
# 2 048 .. 16 384 terms long; as ONE fp32 fma chain (the staged MFMA GEMM) the 4 096-term layer of PointConv cls' second level was the
# row of the network furthest from the fp64 evaluation (sa2 output 6.72 x the 1e-5 bound against an fp32-storage floor of 3.72; round 6:
# 3.61 with chains of 32 terms summed in fp64, csrc/frag.hip).  "auto" (default): chains of 32 where the layer is in the fragment GEMM's
# regime -- K >= 4 096 on <= 8 192 rows (cls sa2: +0.04 ms per step); the 2 048-term layer on 16 384 rows keeps the staged kernel (its
# row moved 5.97 -> 5.75 for +0.05 ms).  PCL_POINTCONV_FLUSH = 0 | 8 | 32 forces one setting on every level (A/B runs).
POINTCONV_LINEAR_FLUSH = os.environ.get("PCL_POINTCONV_FLUSH", "auto")


def _linear_flush(K, rows):
    if POINTCONV_LINEAR_FLUSH != "auto":
        return int(POINTCONV_LINEAR_FLUSH)
    return 32 if 4096 <= K <= 8192 and rows <= 8192 else 0        # (K = 16 384 on the GroupAll level's 32 rows: the wide head kernels, csrc/head.hip)


def _pointconv_linear(linear, out):
    """``self.linear(out)`` with the accumulation chosen for this call's shape (``_linear_flush``)."""
    K = out.shape[-1]
    linear.flush_k = _linear_flush(K, out.numel() // K)
    return linear(out)


def farthest_point_sample(xyz, npoint, start_idx=None):
    """xyz [B,N,3] -> int32 [B,npoint] (:74-116).  ``start_idx`` [B] int32; default: random like :88."""
    B, N, _ = xyz.shape
    if start_idx is None:
        start_idx = torch.randint(0, N, (B,), device=xyz.device, dtype=torch.int32)
    idx, _ = furthest_point_sample(xyz, npoint, tie_stride=1, skip_sqnorm_le=None, start_idx=start_idx)
    return idx


_KNN_POINT_FORM = os.environ.get("PCL_KNN_POINT", "direct")       # "matmul": the reference's own arithmetic (named second definition)


def knn_point(nsample, xyz, new_xyz, form=None):
    """k nearest points of xyz [B,N,3] for each new_xyz [B,S,3] -> int32 [B,S,nsample], ascending distance (:120-131).
    ``form``: "direct" (the library's definition: direct-form squared distances, pcl_knn_f32) or "matmul" (the reference's
    -2ab + a^2 + b^2 with its operation order, pcl_knn_point_matmul_f32, bit-exact against the oracle's restatement of it); default
    from PCL_KNN_POINT."""
    if (form or _KNN_POINT_FORM) == "matmul":
        xyz, new_xyz = _dev(xyz, "xyz"), _dev(new_xyz, "new_xyz")
        B, N, _ = xyz.shape
        S = new_xyz.shape[1]
        out = torch.empty((B, S, nsample), dtype=torch.int32, device=xyz.device)
        _lib.call("pcl_knn_point_matmul_f32", _p(xyz), _p(new_xyz), B, N, S, int(nsample), 1, _p(out), _stream())
        return out
    idx = knn_indices(new_xyz.transpose(1, 2).contiguous(), xyz.transpose(1, 2).contiguous(), nsample)   # [B,k,S]
    return idx.permute(0, 2, 1).contiguous()


def compute_density(xyz, bandwidth):
    """xyz [B,N,3] -> [B,N] (:174-184)."""
    xyz = _dev(xyz, "xyz")
    B, N, _ = xyz.shape
    out = torch.empty((B, N), dtype=torch.float32, device=xyz.device)
    _lib.call("pcl_density_f32", _p(xyz), B, N, float(bandwidth), _p(out), _stream())
    return out


def sample_level(npoint, nsample, xyz, start_idx=None, knn_idx=None):
    """The index-producing half of ``sample_and_group`` (depends on xyz only): FPS centres (:139-140) and their k-NN groups (:141)
    -> (new_xyz [B,S,3], idx int32 [B,S,ns])."""
    fps_idx = farthest_point_sample(xyz, npoint, start_idx)
    new_xyz = index_points(xyz, fps_idx)
    idx = knn_point(nsample, xyz, new_xyz) if knn_idx is None else _dev(knn_idx, "knn_idx", torch.int32)
    return new_xyz, idx


def sample_and_group(npoint, nsample, xyz, points, density_scale=None, start_idx=None, knn_idx=None, sampled=None):
    """(:133-170) -> new_xyz [B,S,3], new_points [B,S,ns,3+D], grouped_xyz_norm [B,S,ns,3], idx, grouped_density.
    ``knn_idx`` int32 [B,S,ns]: neighbour groups to use instead of ``knn_point``'s (parity tests feed the groups of the
    reference's matmul-form arithmetic, :34-53, through here).  ``sampled`` = a ``sample_level`` result produced ahead."""
    B, N, C = xyz.shape
    new_xyz, idx = sampled if sampled is not None else sample_level(npoint, nsample, xyz, start_idx, knn_idx)
    grouped_xyz_norm = index_points(xyz, idx) - new_xyz.view(B, npoint, 1, C)
    if points is not None:
        new_points = torch.cat([grouped_xyz_norm, index_points(points, idx)], dim=-1)
    else:
        new_points = grouped_xyz_norm
    if density_scale is None:
        return new_xyz, new_points, grouped_xyz_norm, idx
    return new_xyz, new_points, grouped_xyz_norm, idx, index_points(density_scale, idx)


def sample_and_group_all(xyz, points, density_scale=None):
    """Missing upstream (:380).  One group of all N points; xyz relative to the centroid; density [B,1,N,1]."""
    B, N, C = xyz.shape
    new_xyz = xyz.mean(dim=1, keepdim=True)
    grouped_xyz = xyz.view(B, 1, N, C) - new_xyz.view(B, 1, 1, C)
    new_points = torch.cat([grouped_xyz, points.view(B, 1, N, -1)], dim=-1) if points is not None else grouped_xyz
    if density_scale is None:
        return new_xyz, new_points, grouped_xyz
    return new_xyz, new_points, grouped_xyz, density_scale.view(B, 1, N, 1)


class _PointConvContract(torch.autograd.Function):
    """out[B,S,C*16] = sum_s feat[B,S,ns,C] * density[B,S,ns,1] * weights[B,S,ns,16]   (:393-394)."""

    @staticmethod
    def forward(ctx, feat, density, weights):
        feat, density, weights = _dev(feat, "feat"), _dev(density, "density"), _dev(weights, "weights")
        B, S, ns, C = feat.shape
        M = weights.shape[-1]
        out = torch.empty((B, S, C * M), dtype=torch.float32, device=feat.device)
        _lib.call("pcl_pointconv_contract_f32", _p(feat), _p(density), _p(weights), B * S, ns, C, M, _p(out), _stream(),
                  algo_bytes=4 * B * S * (ns * (C + 1 + M) + C * M), algo_flops=2 * B * S * ns * C * M)
        ctx.save_for_backward(feat, density, weights)
        return out

    @staticmethod
    def backward(ctx, gout):
        feat, density, weights = ctx.saved_tensors
        B, S, ns, C = feat.shape
        M = weights.shape[-1]
        gout = _dev(gout, "grad")
        dfeat, dw, dd = torch.empty_like(feat), torch.empty_like(weights), torch.empty_like(density)
        _lib.call("pcl_pointconv_contract_bwd_f32", _p(gout), _p(feat), _p(density), _p(weights), B * S, ns, C, M, _p(dfeat),
                  _p(dw), _p(dd), _stream())
        return dfeat, dd, dw


class _PointConvContractBN(torch.autograd.Function):
    """The contraction with the feature MLP's last BatchNorm + activation folded into its feature load (csrc/pointconv.hip,
    FeatBN): ``Y`` [B,S,ns,C] is that layer's pre-BatchNorm output (mlp_hip.stack_plain_deferred), z = lrelu(scale*y + shift) is
    formed on the fly; the backward hands the deferring stack du (masked) and its BatchNorm-backward sums through ``link``.
    Saves one write + one read of the [B,S,ns,C] activation forward and an 805 MB elementwise pass backward per level."""

    @staticmethod
    def forward(ctx, Y, link, density, weights):
        B, S, ns, C = Y.shape
        M = weights.shape[-1]
        density, weights = _dev(density, "density"), _dev(weights, "weights")
        out = torch.empty((B, S, C * M), dtype=torch.float32, device=Y.device)
        _lib.call("pcl_pointconv_contract_bn_f32", _p(Y), _p(link.scale), _p(link.shift), float(link.slope), _p(density), _p(weights), B * S, ns, C,
                  M, _p(out), _stream(), algo_bytes=4 * B * S * (ns * (C + 1 + M) + C * M), algo_flops=2 * B * S * ns * C * M)
        ctx.link = link
        ctx.save_for_backward(Y, density, weights, link.scale, link.shift)
        return out

    @staticmethod
    def backward(ctx, gout):
        Y, density, weights, scale, shift = ctx.saved_tensors
        link = ctx.link
        B, S, ns, C = Y.shape
        M = weights.shape[-1]
        gout = _dev(gout, "grad")
        du, dw, dd = torch.empty_like(Y), torch.empty_like(weights), torch.empty_like(density)
        rows = _lib.size_query("pcl_pointconv_contract_bn_stat_rows", B * S)
        stats = torch.empty((rows, 2, C), dtype=torch.float64, device=Y.device)
        _lib.call("pcl_pointconv_contract_bn_bwd_f32", _p(gout), _p(Y), _p(scale), _p(shift), float(link.slope), _p(density), _p(weights), B * S,
                  ns, C, M, _p(du), _p(dw), _p(dd), _p(stats), _stream())
        link.stats, link.rows = stats, rows
        return du, None, dd, dw


def grouped_feature_mlp_contract(mlp, xyz, new_xyz, points, idx, grouped_density, weights):
    """``feature_mlp_contract(mlp, cat([xyz[idx] - new_xyz, points[idx]]), ...)`` without the grouped tensor: the feature MLP's first
    conv is folded into the grouping (``W0 [dxyz | f] = W0[:, :3] dxyz + (points W0[:, 3:]^T)[idx]``: one GEMM over the N points),
    its last BatchNorm + activation into the contraction.  None when that path does not apply (the caller builds the tensor)."""
    from . import mlp_hip
    r = mlp_hip.stack_grouped_deferred(mlp, xyz, new_xyz, points, idx)
    if r is None:
        return None
    Y, link = r
    B, S, ns = idx.shape
    return _PointConvContractBN.apply(Y.view(B, S, ns, Y.shape[-1]), link, grouped_density.reshape(B, S, ns).contiguous(), weights)


def feature_mlp_contract(mlp, new_points, grouped_density, weights):
    """``pointconv_contract(mlp(new_points), grouped_density, weights)`` (misc/pointconv_utils.py:384-394).  Where the feature MLP
    runs on the per-stack entry points its last BatchNorm + activation are folded into the contraction's feature load."""
    from . import mlp_hip
    B, S, ns, _ = new_points.shape
    if new_points.is_cuda and new_points.dtype == torch.float32 and mlp.last_act is not None:
        r = mlp_hip.stack_plain_deferred(mlp, new_points.reshape(B * S * ns, new_points.shape[-1]).contiguous())
        if r is not None:
            Y, link = r
            return _PointConvContractBN.apply(Y.view(B, S, ns, Y.shape[-1]), link, grouped_density.reshape(B, S, ns).contiguous(), weights)
    return pointconv_contract(mlp(new_points.contiguous()), grouped_density, weights)


def pointconv_contract(new_points, grouped_density, weights):
    """The density multiply + per-point (C x ns)(ns x 16) contraction of PointConv as one HIP kernel:
    new_points [B,S,ns,C], grouped_density [B,S,ns,1], weights [B,S,ns,16] -> [B,S,C*16]."""
    B, S, ns, _ = new_points.shape
    return _PointConvContract.apply(new_points, grouped_density.reshape(B, S, ns), weights)


class DensityNet(nn.Module):
    """Conv1d 1->8->8->1, each + BatchNorm1d + ReLU (:186-218; the sigmoid branch never fires upstream)."""

    def __init__(self, hidden_unit=(8, 8)):
        super().__init__()
        self.mlp = PointwiseMLP([1] + list(hidden_unit) + [1], bias=True)

    def forward(self, xyz_density):
        """[B,N] -> [B,N,1]."""
        return self.mlp(xyz_density.unsqueeze(-1))


class WeightNet(nn.Module):
    """Conv2d in->8->8->out, each + BatchNorm + ReLU (:220-250)."""

    def __init__(self, in_channel, out_channel, hidden_unit=(8, 8)):
        super().__init__()
        self.mlp = PointwiseMLP([in_channel] + list(hidden_unit or []) + [out_channel], bias=True)

    def forward(self, localized_xyz):
        """[B,S,ns,3] -> [B,S,ns,out]."""
        return self.mlp(localized_xyz)


class PointConvDensitySetAbstraction(nn.Module):
    """(:340-400).  ``forward(xyz [B,3,N], points [B,D,N] | None) -> (new_xyz [B,3,S], new_points [B,D',S])``."""

    def __init__(self, npoint, nsample, in_channel, mlp, bandwidth, group_all):
        super().__init__()
        self.npoint = npoint
        self.nsample = nsample
        self.mlp = PointwiseMLP([in_channel] + list(mlp), bias=True)        # mlp_convs + mlp_bns + relu  :348-351
        self.weightnet = WeightNet(3, 16)
        self.densitynet = DensityNet()
        self.linear = PointwiseMLP([16 * mlp[-1], mlp[-1]], bias=True)       # Linear + BatchNorm1d + ReLU  :395-397
        self.group_all = group_all
        self.bandwidth = bandwidth

    def sample(self, xyz, start_idx=None):
        """Everything of this level that depends on the coordinates only (xyz [B,N,3], no gradients): the kernel density (:376), the
        FPS centres and their k-NN groups (:139-141).  -> (new_xyz | None, [(idx | None, density)]) for ``forward(sampling=...)``."""
        density = compute_density(xyz, self.bandwidth)
        if self.group_all:
            return None, [(None, density)]
        new_xyz, idx = sample_level(self.npoint, self.nsample, xyz, start_idx)
        return new_xyz, [(idx, density)]

    def forward(self, xyz, points, start_idx=None, knn_idx=None, sampling=None):
        B, _, N = xyz.shape
        xyz = xyz.permute(0, 2, 1).contiguous()
        if points is not None:
            points = points.permute(0, 2, 1).contiguous()
        sampled = None
        if sampling is not None:                                                         # produced ahead by ``sample`` (e.g. on a side stream)
            (idx_pre, density_pre) = sampling[1][0]
            sampled = None if idx_pre is None else (sampling[0], idx_pre)
        density_scale = self.densitynet(density_pre if sampling is not None else compute_density(xyz, self.bandwidth))   # :376-377  [B,N,1]
        if not self.group_all and points is not None and xyz.is_cuda:
            # sample_and_group :133-170 WITHOUT the [B,S,ns,3+D] tensor: indices, centres, local coordinates and gathered density
            # only; the feature MLP's first conv runs folded into the grouping (grouped_feature_mlp_contract)
            new_xyz, idx = sampled if sampled is not None else sample_level(self.npoint, self.nsample, xyz, start_idx, knn_idx)
            grouped_xyz_norm = index_points(xyz, idx) - new_xyz.view(B, self.npoint, 1, 3)
            grouped_density = index_points(density_scale, idx)
            weights = self.weightnet(grouped_xyz_norm.contiguous())                      # [B,S,ns,16]  :391-392
            out = grouped_feature_mlp_contract(self.mlp, xyz, new_xyz, points, idx, grouped_density, weights)
            if out is None:
                new_points = torch.cat([grouped_xyz_norm, index_points(points, idx)], dim=-1)
                out = feature_mlp_contract(self.mlp, new_points, grouped_density, weights)
            new_points = _pointconv_linear(self.linear, out)                              # :395-397
            return new_xyz.permute(0, 2, 1), new_points.permute(0, 2, 1)
        if self.group_all:
            new_xyz, new_points, grouped_xyz_norm, grouped_density = sample_and_group_all(xyz, points, density_scale)
        else:
            new_xyz, new_points, grouped_xyz_norm, _, grouped_density = sample_and_group(
                self.npoint, self.nsample, xyz, points, density_scale, start_idx, knn_idx, sampled)
        weights = self.weightnet(grouped_xyz_norm.contiguous())                          # [B,S,ns,16]  :391-392
        # feature MLP :384-389 + density multiply and per-point matmul :393-394 (the MLP's last BatchNorm + ReLU ride in the contraction)
        new_points = feature_mlp_contract(self.mlp, new_points, grouped_density, weights)     # [B,S,C*16]
        new_points = _pointconv_linear(self.linear, new_points)                           # :395-397
        return new_xyz.permute(0, 2, 1), new_points.permute(0, 2, 1)

    def execute(self, *a, **k):
        return self(*a, **k)


class PointConvDensitySetInterpolation(nn.Module):
    """(:252-330).  ``forward(xyz1 [B,3,N], xyz2 [B,3,S], points1 [B,D,N], points2 [B,D2,S]) -> [B,D',N]``.

    3-NN inverse-distance interpolation of ``points2`` onto ``xyz1`` (HIP three_nn + three_interpolate; the
    reference sorts a dense [B,N,S] matrix, :293-301), then a PointConv over ``xyz1`` itself.  Kept as written
    upstream: ``points1`` is accepted and ignored (:286,:305), and the grouping step is
    ``sample_and_group(N, nsample, ...)`` -- i.e. FPS asked for all N points (:305), so the output rows follow the
    FPS visiting order (random start) rather than the order of ``xyz1``."""

    def __init__(self, nsample, in_channel, mlp, bandwidth):
        super().__init__()
        self.bandwidth = bandwidth
        self.nsample = nsample
        self.in_channel = in_channel
        self.mlp = PointwiseMLP([in_channel] + list(mlp), bias=True)
        self.weightnet = WeightNet(3, 16)
        self.densitynet = DensityNet()
        self.linear = PointwiseMLP([16 * mlp[-1], mlp[-1]], bias=True)

    def forward(self, xyz1, xyz2, points1, points2, start_idx=None):
        xyz1 = xyz1.permute(0, 2, 1).contiguous()
        xyz2 = xyz2.permute(0, 2, 1).contiguous()
        points2 = points2.permute(0, 2, 1).contiguous()
        B, N, _ = xyz1.shape
        idx3, w3 = three_nn(xyz1, xyz2)                                                   # :293-299
        interpolated = three_interpolate(points2, idx3, w3)                               # :300
        density_scale = self.densitynet(compute_density(xyz1, self.bandwidth))            # :304-305
        if xyz1.is_cuda:
            # sample_and_group(N, nsample, ...) :307 without the [B,N,ns,3+D] tensor, as in PointConvDensitySetAbstraction.forward:
            # the feature MLP's first conv runs folded into the grouping (K = 3+D GEMM over the N points, not over N*ns rows)
            fps_idx = farthest_point_sample(xyz1, N, start_idx)
            new_xyz = index_points(xyz1, fps_idx)
            idx = knn_point(self.nsample, xyz1, new_xyz)
            grouped_xyz_norm = index_points(xyz1, idx) - new_xyz.view(B, N, 1, 3)
            grouped_density = index_points(density_scale, idx)
            weights = self.weightnet(grouped_xyz_norm.contiguous())                       # :317-318
            out = grouped_feature_mlp_contract(self.mlp, xyz1, new_xyz, interpolated, idx, grouped_density, weights)
            if out is None:
                new_points = torch.cat([grouped_xyz_norm, index_points(interpolated, idx)], dim=-1)
                out = feature_mlp_contract(self.mlp, new_points, grouped_density, weights)
            return _pointconv_linear(self.linear, out).permute(0, 2, 1)                   # :321-323
        _, new_points, grouped_xyz_norm, _, grouped_density = sample_and_group(
            N, self.nsample, xyz1, interpolated, density_scale, start_idx)                # :307
        weights = self.weightnet(grouped_xyz_norm.contiguous())                           # :317-318
        new_points = feature_mlp_contract(self.mlp, new_points, grouped_density, weights)     # :311-315 + :319-320
        return _pointconv_linear(self.linear, new_points).permute(0, 2, 1)                # :321-323

    def execute(self, *a, **k):
        return self(*a, **k)
