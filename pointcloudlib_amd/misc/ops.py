"""Point-cloud operators -- host-side mirror of the reference's ``misc/ops.py``.

Same class names, constructor arguments, call arguments, shapes, dtypes and channel-last layouts as
/root/reference/misc/ops.py (FurthestPointSampler :114, BallQueryGrouper :289, GroupAll :410,
KNN :422, PointNetFeaturePropagation :54, index_points :12), written as ``torch.nn.Module``s with an
``execute`` alias for Jittor's method name.  All arithmetic runs in ``libpcl_hip.so`` (hand-written
gfx950 kernels) through the C ABI of ``include/pcl_hip.h``; PyTorch only owns the device buffers, the
stream and autograd bookkeeping.  CPU tensors are rejected: there is no fallback path.
"""
import math

import os

import torch
from torch import nn

from .. import _lib

__all__ = [
    "optimal_block", "furthest_point_sample", "ball_query", "ball_query_multi", "group_offsets_multi", "group_points", "group_points_compact", "RowSet", "group_all",
    "index_points",
    "knn_indices", "edge_features", "three_nn", "three_interpolate", "FurthestPointSampler", "BallQueryGrouper", "GroupAll",
    "KNN", "PointNetFeaturePropagation",
]


# ----------------------------------------------------------------------------- plumbing
def _dev(t, name, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor on the GPU (libpcl_hip has no CPU path), got {t.device}")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t.contiguous()


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    # the raw hipStream_t of torch's current stream (torch.cuda.current_stream().cuda_stream costs ~10 us of Python per call)
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def optimal_block(batch_size):
    """misc/ops.py:110-111 -- ``2 ** int(math.log(batch_size))`` (natural log, as written)."""
    return 2 ** int(math.log(batch_size)) if batch_size >= 1 else 1


# ----------------------------------------------------------------------------- index producers
def furthest_point_sample(xyz, n_samples, tie_stride=None, skip_sqnorm_le=1e-3, start_idx=None):
    """xyz [B,N,3] f32 -> (idx [B,n] int32, new_xyz [B,n,3]).  misc/ops.py:124-234, :280-284.

    ``tie_stride`` defaults to the reference's launch block size ``optimal_block(B)``, which fixes how
    exact distance ties are broken; ``skip_sqnorm_le=None`` disables the near-origin skip
    (misc/pointconv_utils.py:74-116 has none)."""
    xyz = _dev(xyz, "xyz")
    if xyz.dim() != 3 or xyz.shape[2] != 3:
        raise ValueError(f"xyz must be [B,N,3], got {tuple(xyz.shape)}")
    B, N, _ = xyz.shape
    if not (1 <= n_samples <= N):
        raise ValueError(f"n_samples={n_samples} must be in [1, N={N}]")  # assert at misc/ops.py:269
    if tie_stride is None:
        tie_stride = optimal_block(B)
    start_idx = _dev(start_idx, "start_idx", torch.int32)
    idx = torch.empty((B, n_samples), dtype=torch.int32, device=xyz.device)
    new_xyz = torch.empty((B, n_samples, 3), dtype=torch.float32, device=xyz.device)
    thr = -1.0 if skip_sqnorm_le is None else float(skip_sqnorm_le)
    _lib.call("pcl_fps_f32", _p(xyz), B, N, n_samples, int(tie_stride), thr, _p(start_idx), _p(idx),
                                      _p(new_xyz), _stream(), algo_bytes=B * (12 * N + 16 * n_samples))
    return idx, new_xyz


def ball_query(new_xyz, xyz, radius, n_samples, return_cnt=False):
    """new_xyz [B,m,3], xyz [B,N,3] -> idx [B,m,ns] int32.  misc/ops.py:291-330."""
    new_xyz = _dev(new_xyz, "new_xyz")
    xyz = _dev(xyz, "xyz")
    if new_xyz.dim() != 3 or new_xyz.shape[2] != 3 or xyz.dim() != 3 or xyz.shape[2] != 3:
        raise ValueError("new_xyz / xyz must be [B,*,3]")
    if new_xyz.shape[0] != xyz.shape[0]:
        raise ValueError("batch size mismatch")   # assert at misc/ops.py:361
    B, m, _ = new_xyz.shape
    N = xyz.shape[1]
    idx = torch.empty((B, m, n_samples), dtype=torch.int32, device=xyz.device)
    cnt = torch.empty((B, m), dtype=torch.int32, device=xyz.device) if return_cnt else None
    _lib.call("pcl_ball_query_f32", _p(new_xyz), _p(xyz), B, m, N, float(radius), int(n_samples), _p(idx),
                                             _p(cnt), _stream(), algo_bytes=B * (12 * (N + m) + 4 * m * n_samples))
    return (idx, cnt) if return_cnt else idx


BALL_QUERY_MULTI_MAX = 4          # radii per pcl_ball_query_multi_f32 call


def ball_query_multi(new_xyz, xyz, radii, n_samples, return_cnt=False):
    """Ball queries of several radii around the same centres in ONE scan of the cloud (multi-scale grouping: one BallQueryGrouper per
    scale on the same new_xyz, reference networks/seg/pointnet2_partseg.py:93-103).  -> [idx [B,m,ns_r]] or [(idx, cnt)] per radius,
    each identical to ``ball_query(new_xyz, xyz, radii[r], n_samples[r])``."""
    import ctypes
    new_xyz = _dev(new_xyz, "new_xyz")
    xyz = _dev(xyz, "xyz")
    if new_xyz.dim() != 3 or new_xyz.shape[2] != 3 or xyz.dim() != 3 or xyz.shape[2] != 3:
        raise ValueError("new_xyz / xyz must be [B,*,3]")
    if new_xyz.shape[0] != xyz.shape[0]:
        raise ValueError("batch size mismatch")
    n = len(radii)
    if n != len(n_samples) or not 1 <= n <= BALL_QUERY_MULTI_MAX:
        raise ValueError(f"{n} radii / {len(n_samples)} sample counts: 1..{BALL_QUERY_MULTI_MAX} of each")
    B, m, _ = new_xyz.shape
    N = xyz.shape[1]
    idx = [torch.empty((B, m, int(s)), dtype=torch.int32, device=xyz.device) for s in n_samples]
    cnt = [torch.empty((B, m), dtype=torch.int32, device=xyz.device) for _ in range(n)] if return_cnt else None
    c_r = (ctypes.c_float * n)(*[float(r) for r in radii])
    c_s = (ctypes.c_int32 * n)(*[int(s) for s in n_samples])
    c_i = (ctypes.c_void_p * n)(*[t.data_ptr() for t in idx])
    c_c = (ctypes.c_void_p * n)(*[t.data_ptr() for t in cnt]) if return_cnt else None
    _lib.call("pcl_ball_query_multi_f32", _p(new_xyz), _p(xyz), B, m, N, n, c_r, c_s, c_i, c_c, _stream(),
              algo_bytes=B * (12 * (N + m) + 4 * m * sum(int(s) for s in n_samples)))
    return list(zip(idx, cnt)) if return_cnt else idx


# The k-NN distance has two definitions in this library: the default rounds `tmp*tmp` and the sum
# separately, as the reference's source text reads (misc/ops.py:488-491); "fma" is the NAMED second definition -- what nvcc's
# default -fmad=true makes of that line -- one VALU operation fewer per element.  Opt-in: PCL_KNN_CONTRACT=fma or
# ``ops.KNN_CONTRACT = "fma"``; each is bit-exact against the oracle evaluated under the same reading.
KNN_CONTRACT = os.environ.get("PCL_KNN_CONTRACT", "")


def knn_indices(x_q, x_r, k, contract=None):
    """KNN(k).execute(x_q [B,C,Nq], x_r [B,C,Nr]) -> int32 [B,k,Nq].  misc/ops.py:651-663."""
    contract = KNN_CONTRACT if contract is None else contract
    if contract not in ("", "fma"):
        raise ValueError(f"knn contract {contract!r}: '' (source reading) or 'fma'")
    x_q = _dev(x_q, "x_q")
    x_r = _dev(x_r, "x_r")
    if x_q.dim() != 3 or x_r.dim() != 3 or x_q.shape[:2] != x_r.shape[:2]:
        raise ValueError(f"x_q/x_r must be [B,C,N*] with equal B and C, got {tuple(x_q.shape)} {tuple(x_r.shape)}")
    B, C, Nq = x_q.shape
    Nr = x_r.shape[2]
    idx = torch.empty((B, k, Nq), dtype=torch.int32, device=x_q.device)
    nbytes = _lib.lib().pcl_knn_workspace_bytes(B, C, Nr, Nq, k)
    ws = torch.empty((max(nbytes, 4) + 3) // 4, dtype=torch.float32, device=x_q.device)   # the reference's tmp_dist
    _lib.call("pcl_knn_fma_f32" if contract == "fma" else "pcl_knn_f32", _p(x_r), _p(x_q), B, C, Nr, Nq, int(k), _p(idx), _p(ws), nbytes,
              _stream(), algo_bytes=4 * B * C * (Nr + Nq) + 4 * B * k * Nq, algo_flops=3 * B * Nr * Nq * C)
    return idx


_KNN_NK = os.environ.get("PCL_KNN_NK", "1") != "0"          # lab switch (A/B on one box): 0 = the reference-layout search + a permute copy


def knn_lists(x_q, x_r, k):
    """``knn_indices(x_q, x_r, k).permute(0, 2, 1).contiguous()`` -- the neighbour lists as [B, Nq, k] rows (what ``get_graph_feature``
    makes of KNN's result, networks/cls/dgcnn.py:34-35) -- written in that layout by the search itself where the fused kernel applies."""
    if _KNN_NK and KNN_CONTRACT == "" and x_r.is_cuda and _lib.lib().pcl_knn_nk_supported(x_r.shape[2]):
        x_q, x_r = _dev(x_q, "x_q"), _dev(x_r, "x_r")
        B, C, Nq = x_q.shape
        Nr = x_r.shape[2]
        idx = torch.empty((B, Nq, k), dtype=torch.int32, device=x_q.device)
        _lib.call("pcl_knn_nk_f32", _p(x_r), _p(x_q), B, C, Nr, Nq, int(k), _p(idx), _stream(),
                  algo_bytes=4 * B * C * (Nr + Nq) + 4 * B * k * Nq, algo_flops=3 * B * Nr * Nq * C)
        return idx
    return knn_indices(x_q, x_r, k).permute(0, 2, 1).contiguous()


def three_nn(xyz1, xyz2):
    """xyz1 [B,N,3] targets, xyz2 [B,S,3] sources -> (idx [B,N,3] int32, weight [B,N,3])."""
    xyz1 = _dev(xyz1, "xyz1")
    xyz2 = _dev(xyz2, "xyz2")
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    idx = torch.empty((B, N, 3), dtype=torch.int32, device=xyz1.device)
    w = torch.empty((B, N, 3), dtype=torch.float32, device=xyz1.device)
    _lib.call("pcl_three_nn_f32", _p(xyz1), _p(xyz2), B, N, S, _p(idx), _p(w), _stream())
    return idx, w


# ----------------------------------------------------------------------------- differentiable gathers
class _GroupPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, new_xyz, feat, idx, use_xyz):
        idx = _dev(idx, "idx", torch.int32)
        B, m, ns = idx.shape
        xyz = _dev(xyz, "xyz")
        new_xyz = _dev(new_xyz, "new_xyz")
        feat = _dev(feat, "feature")
        N = xyz.shape[1]
        C = 0 if feat is None else feat.shape[2]
        D = (3 if use_xyz else 0) + C
        out = torch.empty((B, m, ns, D), dtype=torch.float32, device=idx.device)
        _lib.call("pcl_group_f32", _p(xyz), _p(new_xyz), _p(feat), _p(idx), B, N, m, ns, C, int(use_xyz),
                                            _p(out), _stream(),
                  algo_bytes=B * (4 * m * ns + 12 * m + 4 * N * (3 + C) + 4 * m * ns * D))
        ctx.save_for_backward(idx)
        ctx.dims = (B, N, m, ns, C, int(use_xyz))
        return out

    @staticmethod
    def backward(ctx, gout):
        (idx,) = ctx.saved_tensors
        B, N, m, ns, C, use_xyz = ctx.dims
        gfeat = None
        if C > 0 and ctx.needs_input_grad[2]:
            gout = _dev(gout, "grad")
            gfeat = torch.empty((B, N, C), dtype=torch.float32, device=gout.device)
            _lib.call("pcl_group_bwd_f32", _p(gout), _p(idx), B, N, m, ns, C, use_xyz, _p(gfeat), _stream(),
                      algo_bytes=B * (4 * m * ns * C + 4 * m * ns + 4 * N * C))
        return None, None, gfeat, None, None


def group_points(xyz, new_xyz, feature, idx, use_xyz=True):
    """[B,m,ns,(3)+C] = concat(xyz[idx]-new_xyz, feature[idx]).  misc/ops.py:383-407."""
    if not use_xyz and feature is None:
        raise ValueError("use_xyz=False needs features")
    return _GroupPoints.apply(xyz, new_xyz, feature, idx, bool(use_xyz))


_GROUP_ALL_BWD_VIEW = __import__("os").environ.get("PCL_GROUP_ALL_BWD_VIEW", "1") != "0"      # lab switch (A/B on one box): 0 = the copy kernel


class _GroupAll(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, feat, use_xyz):
        xyz = _dev(xyz, "xyz")
        feat = _dev(feat, "feature")
        B, N, _ = xyz.shape
        C = 0 if feat is None else feat.shape[2]
        D = (3 if use_xyz else 0) + C
        out = torch.empty((B, 1, N, D), dtype=torch.float32, device=xyz.device)
        _lib.call("pcl_group_all_f32", _p(xyz), _p(feat), B, N, C, int(use_xyz), _p(out), _stream())
        ctx.dims = (B, N, C, int(use_xyz))
        return out

    @staticmethod
    def backward(ctx, gout):
        B, N, C, use_xyz = ctx.dims
        gfeat = None
        if C > 0 and ctx.needs_input_grad[1]:
            D = (3 if use_xyz else 0) + C
            if _GROUP_ALL_BWD_VIEW and gout.is_contiguous() and gout.is_cuda and gout.dtype == torch.float32:
                # the feature columns of the gradient AS A VIEW (row stride D): the stack below reads its gout in place through gout_ld
                # (csrc/stack.hip), any other consumer makes it dense itself -- no copy launch here (round 6)
                return None, gout.view(B, N, D)[:, :, D - C:], None
            gout = _dev(gout, "grad")
            gfeat = torch.empty((B, N, C), dtype=torch.float32, device=gout.device)
            _lib.call("pcl_group_all_bwd_f32", _p(gout), B, N, C, use_xyz, _p(gfeat), _stream())
        return None, gfeat, None


class RowSet:
    """Metadata of duplicate-compacted grouped rows (see include/pcl_hip.h, 'ragged groups')."""

    def __init__(self, B, m, ns, row_meta, row_src, group_off):
        self.B, self.m, self.ns = B, m, ns
        self.G = B * m
        self.row_meta, self.row_src, self.group_off = row_meta, row_src, group_off
        self.n_rows_dev = group_off[self.G:]            # 1-element view: the valid-row count stays on the device

    @property
    def capacity(self):
        return self.G * self.ns


def group_offsets(cnt):
    """cnt int32 [B,m] (ball-query hit counts) -> group_off int32 [B*m+1]: exclusive scan of max(cnt,1); the last entry
    is the number of distinct rows.  Depends on the indices only, so it can be produced with them (side stream)."""
    cnt = _dev(cnt, "cnt", torch.int32)
    G = cnt.numel()
    group_off = torch.empty((G + 1,), dtype=torch.int32, device=cnt.device)
    _lib.call("pcl_group_offsets_i32", _p(cnt), G, _p(group_off), _stream())
    return group_off


def group_offsets_multi(cnts):
    """``group_offsets`` of up to four count arrays of one size in one launch (the scales of a multi-scale level)."""
    import ctypes
    cnts = [_dev(c, "cnt", torch.int32) for c in cnts]
    n, G = len(cnts), cnts[0].numel()
    if not 1 <= n <= BALL_QUERY_MULTI_MAX or any(c.numel() != G for c in cnts):
        raise ValueError("group_offsets_multi: 1..4 count arrays of equal size")
    offs = [torch.empty((G + 1,), dtype=torch.int32, device=cnts[0].device) for _ in range(n)]
    c_c = (ctypes.c_void_p * n)(*[t.data_ptr() for t in cnts])
    c_o = (ctypes.c_void_p * n)(*[t.data_ptr() for t in offs])
    _lib.call("pcl_group_offsets_multi_i32", n, c_c, G, c_o, _stream())
    return offs


class _GroupCompact(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, new_xyz, feat, idx, cnt, group_off, use_xyz, pad_to):
        idx = _dev(idx, "idx", torch.int32)
        cnt = _dev(cnt, "cnt", torch.int32)
        group_off = _dev(group_off, "group_off", torch.int32)
        B, m, ns = idx.shape
        xyz = _dev(xyz, "xyz")
        new_xyz = _dev(new_xyz, "new_xyz")
        feat = _dev(feat, "feature")
        N = xyz.shape[1]
        C = 0 if feat is None else feat.shape[2]
        D = (3 if use_xyz else 0) + C
        S = (D + pad_to - 1) // pad_to * pad_to            # row stride: zero columns up to a multiple of pad_to
        cap = B * m * ns
        dev = idx.device
        rows = torch.empty((cap, S), dtype=torch.float32, device=dev)
        row_meta = torch.empty((cap, 2), dtype=torch.int32, device=dev)
        row_src = torch.empty((cap,), dtype=torch.int32, device=dev)
        _lib.call("pcl_group_compact_f32", _p(xyz), _p(new_xyz), _p(feat), _p(idx), _p(cnt), _p(group_off), B, N, m, ns, C,
                  int(use_xyz), S, _p(rows), _p(row_meta), _p(row_src), _stream())
        ctx.dims = (B, N, C, S, int(use_xyz), m, ns)
        ctx.mark_non_differentiable(row_meta, row_src)
        ctx.save_for_backward(row_src, group_off)        # saved properly, never as ctx attributes (no cycles)
        return rows, row_meta, row_src

    @staticmethod
    def backward(ctx, grows, *_):
        B, N, C, S, use_xyz, m, ns = ctx.dims
        gfeat = None
        if C > 0 and ctx.needs_input_grad[2]:
            row_src, group_off = ctx.saved_tensors
            grows = _dev(grows, "grad")
            gfeat = torch.empty((B, N, C), dtype=torch.float32, device=grows.device)
            _lib.call("pcl_scatter_rows_add_f32", _p(grows), _p(row_src), _p(group_off[B * m:]), B * m * ns, S,
                      3 if use_xyz else 0, C, B * N, _p(gfeat), _stream())
        return None, None, gfeat, None, None, None, None, None


def group_points_compact(xyz, new_xyz, feature, idx, cnt, use_xyz=True, group_off=None, pad_to=4):
    """Duplicate-compacted grouping: (rows [B*m*ns (capacity), S], RowSet).  Only the first ``group_off[-1]`` rows are
    valid: the DISTINCT points of every ball-query group in (group, slot) order, each with its multiplicity.  Rows are
    ``[xyz - centre | features]`` padded with zero columns to a multiple of ``pad_to`` (S >= 3+C) so that every consumer
    moves them as 16-byte pieces; ``PointwiseMLP`` pads its first weight matrix to match."""
    if group_off is None:
        group_off = group_offsets(cnt)
    rows, row_meta, row_src = _GroupCompact.apply(xyz, new_xyz, feature, idx, cnt, group_off, bool(use_xyz), int(pad_to))
    B, m, ns = idx.shape
    return rows, RowSet(B, m, ns, row_meta, row_src, group_off)


def group_all(xyz, feature, use_xyz=True):
    """[B,1,N,3+C] = concat(xyz, feature) (xyz not re-centred).  misc/ops.py:415-419."""
    return _GroupAll.apply(xyz, feature, bool(use_xyz))


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, idx):
        src = _dev(src, "points")
        idx = _dev(idx, "idx", torch.int32)
        B, N, C = src.shape
        M = idx.numel() // B
        out = torch.empty(tuple(idx.shape) + (C,), dtype=torch.float32, device=src.device)
        _lib.call("pcl_gather_rows_f32", _p(src), _p(idx), B, N, M, C, _p(out), _stream())
        ctx.save_for_backward(idx)
        ctx.dims = (B, N, M, C)
        return out

    @staticmethod
    def backward(ctx, gout):
        (idx,) = ctx.saved_tensors
        B, N, M, C = ctx.dims
        gout = _dev(gout, "grad")
        gsrc = torch.empty((B, N, C), dtype=torch.float32, device=gout.device)
        _lib.call("pcl_gather_rows_bwd_f32", _p(gout), _p(idx), B, N, M, C, _p(gsrc), _stream())
        return gsrc, None


def index_points(points, idx):
    """points [B,N,C], idx [B,S] or [B,S,K] (int32) -> [B,S,(K,)C].  misc/ops.py:12-27."""
    return _GatherRows.apply(points, idx.to(torch.int32))


class _EdgeFeature(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx):
        x = _dev(x, "x")
        idx = _dev(idx, "idx", torch.int32)
        B, N, C = x.shape
        k = idx.shape[2]
        out = torch.empty((B, N, k, 2 * C), dtype=torch.float32, device=x.device)
        _lib.call("pcl_edge_feature_f32", _p(x), _p(idx), B, N, k, C, _p(out), _stream(),
                  algo_bytes=4 * B * N * (C + k + 2 * C * k))
        ctx.save_for_backward(idx)
        ctx.dims = (B, N, k, C)
        return out

    @staticmethod
    def backward(ctx, gout):
        (idx,) = ctx.saved_tensors
        B, N, k, C = ctx.dims
        gout = _dev(gout, "grad")
        gx = torch.empty((B, N, C), dtype=torch.float32, device=gout.device)
        _lib.call("pcl_edge_feature_bwd_f32", _p(gout), _p(idx), B, N, k, C, _p(gx), _stream(),
                  algo_bytes=4 * B * N * (C + k + 2 * C * k))
        return gx, None


def edge_features(x, idx):
    """x [B,N,C] channel-last, idx [B,N,k] int32 -> [B,N,k,2C] = concat(x[idx]-x, x).  dgcnn.py:29-50."""
    return _EdgeFeature.apply(x, idx)


class _ThreeInterpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points2, idx3, w3):
        points2 = _dev(points2, "points2")
        idx3 = _dev(idx3, "idx3", torch.int32)
        w3 = _dev(w3, "w3")
        B, S, D = points2.shape
        N = idx3.shape[1]
        out = torch.empty((B, N, D), dtype=torch.float32, device=points2.device)
        _lib.call("pcl_three_interp_f32", _p(points2), _p(idx3), _p(w3), B, N, S, D, _p(out), _stream())
        ctx.save_for_backward(idx3, w3)
        ctx.dims = (B, N, S, D)
        return out

    @staticmethod
    def backward(ctx, gout):
        idx3, w3 = ctx.saved_tensors
        B, N, S, D = ctx.dims
        gout = _dev(gout, "grad")
        g = torch.empty((B, S, D), dtype=torch.float32, device=gout.device)
        _lib.call("pcl_three_interp_bwd_f32", _p(gout), _p(idx3), _p(w3), B, N, S, D, _p(g), _stream())
        return g, None, None


def three_interpolate(points2, idx3, w3):
    """sum_j w3[b,n,j] * points2[b, idx3[b,n,j], :]  ->  [B,N,D].  misc/ops.py:93."""
    return _ThreeInterpolate.apply(points2, idx3, w3)


# ----------------------------------------------------------------------------- modules (reference names)
class _Module(nn.Module):
    def execute(self, *a, **k):   # Jittor's name for forward
        return self(*a, **k)


class FurthestPointSampler(_Module):
    """misc/ops.py:114-286: ``FurthestPointSampler(n_samples)(x[B,N,3]) -> [B,n_samples,3]``."""

    def __init__(self, n_samples, tie_stride=None):
        super().__init__()
        self.n_samples = n_samples
        self.tie_stride = tie_stride

    def forward(self, x, return_idx=False):
        idx, y = furthest_point_sample(x, self.n_samples, self.tie_stride)
        return (y, idx) if return_idx else y


class BallQueryGrouper(_Module):
    """misc/ops.py:289-407: ``(new_xyz[B,m,3], pointset[B,N,3], feature[B,N,C]|None) -> [B,m,ns,3+C]``."""

    def __init__(self, radius, n_samples, use_xyz):
        super().__init__()
        self.radius = radius
        self.n_samples = n_samples
        self.use_xyz = use_xyz

    def forward(self, new_xyz, pointset, feature, return_idx=False):
        idx = ball_query(new_xyz, pointset, self.radius, self.n_samples)
        if self.use_xyz or feature is not None:
            out = group_points(pointset, new_xyz, feature, idx, self.use_xyz)
        else:
            out = None   # misc/ops.py:405: use_xyz=False and feature=None returns None
        return (out, idx) if return_idx else out


class GroupAll(_Module):
    """misc/ops.py:410-419 (``use_xyz=False`` is a latent NameError upstream; here it returns features only)."""

    def __init__(self, use_xyz):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, new_xyz, pointset, feature):
        return group_all(pointset, feature, self.use_xyz)


class KNN(_Module):
    """misc/ops.py:422-663: ``KNN(k)(x_q[B,C,Nq], x_r[B,C,Nr]) -> int32 [B,k,Nq]``."""

    def __init__(self, k):
        super().__init__()
        self.k = k

    def forward(self, x_q, x_r):
        return knn_indices(x_q, x_r, self.k)


class PointNetFeaturePropagation(_Module):
    """misc/ops.py:54-107.  ``(xyz1[B,N,3], xyz2[B,S,3], points1[B,N,D1]|None, points2[B,S,D2]) -> [B,N,mlp[-1]]``.

    3-NN inverse-distance interpolation (HIP three_nn + three_interpolate instead of the reference's
    dense matrix + full argsort), concat, then Conv1d(k=1, bias)+BatchNorm1d+ReLU per ``mlp`` entry."""

    def __init__(self, in_channel, mlp):
        super().__init__()
        from .layers import PointwiseMLP
        self.mlp = PointwiseMLP([in_channel] + list(mlp), bias=True)

    def forward(self, xyz1, xyz2, points1, points2):
        B, N, _ = xyz1.shape
        S = xyz2.shape[1]
        if S == 1:
            interpolated = points2.expand(B, N, points2.shape[2])          # :83-84
        else:
            idx, w = three_nn(xyz1, xyz2)
            interpolated = three_interpolate(points2, idx, w)              # :86-93
        new_points = torch.cat([points1, interpolated], dim=-1) if points1 is not None else interpolated
        return self.mlp(new_points.contiguous())
