"""The FC classification head on a handful of rows (one per cloud) -- host side of csrc/head.hip.

The counterpart networks keep the reference's module tree (``nn.Linear`` / ``nn.BatchNorm1d`` / ``nn.ReLU`` /
``nn.Dropout``; networks/cls/pointnet2.py:138-147, dgcnn.py:87-93, pointnet.py:22-38) as parameter containers; on the GPU
``head_layer`` / ``fc_head`` run Linear + BatchNorm1d + activation as one kernel per layer (forward) and two (backward)
instead of ~15 library launches per layer pair.  More than 64 rows fall back to the PyTorch modules (plain library GEMMs).
"""
import torch
import torch.nn.functional as F
from torch import nn

from .. import _lib, syncbn
from .ops import _p, _stream

MAX_ROWS = 64


class _HeadLayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b, gamma, beta, rmean, rvar, cfg):
        bn_mode, eps, momentum, slope = cfg
        x = x.contiguous()
        R, K = x.shape
        N = W.shape[0]
        dev = x.device
        ypre = torch.empty((R, N), device=dev)
        out = torch.empty((R, N), device=dev)
        mean = invstd = None
        if bn_mode:
            mean, invstd = torch.empty((N,), device=dev), torch.empty((N,), device=dev)
        _lib.call("pcl_head_layer_fwd_f32", _p(x), _p(W), _p(b), _p(gamma), _p(beta), _p(rmean), _p(rvar), R, K, N, bn_mode, eps,
                  momentum, slope, _p(ypre), _p(out), _p(mean), _p(invstd), _stream())
        ctx.cfg = (bn_mode, slope, b is not None, gamma is not None)
        ctx.save_for_backward(x, W, out, ypre, gamma, mean, invstd)
        return out

    @staticmethod
    def backward(ctx, g):
        bn_mode, slope, has_b, has_g = ctx.cfg
        x, W, out, ypre, gamma, mean, invstd = ctx.saved_tensors
        R, K = x.shape
        N = W.shape[0]
        dev = x.device
        g = g.contiguous()
        dY = torch.empty((R, N), device=dev)
        dW = torch.empty_like(W)
        db = torch.empty((N,), device=dev) if has_b else None
        dgamma = torch.empty((N,), device=dev) if (bn_mode and has_g) else None
        dbeta = torch.empty((N,), device=dev) if (bn_mode and has_g) else None
        dX = torch.empty((R, K), device=dev) if ctx.needs_input_grad[0] else None
        _lib.call("pcl_head_layer_bwd_f32", _p(x), _p(W), _p(g), _p(out), _p(ypre), _p(gamma), _p(mean), _p(invstd), R, K, N,
                  bn_mode, slope, _p(dY), _p(dW), _p(db), _p(dgamma), _p(dbeta), _p(dX), _stream())
        return dX, dW, db, dgamma, dbeta, None, None, None


def _slope_of(act):
    if act is None:
        return 1.0
    if isinstance(act, nn.ReLU):
        return 0.0
    if isinstance(act, nn.LeakyReLU):
        return float(act.negative_slope)
    raise TypeError(f"unsupported head activation {type(act).__name__}")


def head_layer(x, linear, bn=None, act=None):
    """``act(bn(linear(x)))`` for x [R, K]; ``act``: None, ``nn.ReLU``/``nn.LeakyReLU`` instance or a negative slope."""
    slope = float(act) if isinstance(act, (int, float)) else _slope_of(act)
    if not x.is_cuda:
        raise RuntimeError("head_layer: expected a tensor on the GPU (libpcl_hip has no CPU path)")
    if x.shape[0] > MAX_ROWS or x.dim() != 2 or x.dtype != torch.float32:     # large batches: plain library GEMMs
        y = linear(x)
        if bn is not None:
            y = syncbn.batch_norm_1d(y, bn) if (bn.training and syncbn.active() and y.dim() == 2) else bn(y)
        return y if slope == 1.0 else F.leaky_relu(y, slope)
    if bn is not None and bn.training and syncbn.active():
        # synchronised statistics over all ranks' rows: the one-kernel layer computes its batch statistics inside the kernel,
        # so this case (R <= 64 rows per rank) runs the library GEMM + syncbn.batch_norm_1d (fp64 sums, biased running variance)
        y = syncbn.batch_norm_1d(linear(x), bn)
        return y if slope == 1.0 else F.leaky_relu(y, slope)
    if bn is None:
        cfg = (0, 0.0, 0.0, slope)
        return _HeadLayer.apply(x, linear.weight, linear.bias, None, None, None, None, cfg)
    # | 4: running_var follows Jittor's rule (biased batch variance) like PointwiseMLP / pcl_bn_finalize_f32
    mode = (1 if bn.training else 2) | 4
    momentum = 0.1 if bn.momentum is None else bn.momentum
    cfg = (mode, bn.eps, momentum, slope)
    return _HeadLayer.apply(x, linear.weight, linear.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, cfg)


def fc_head(seq, x):
    """Run an ``nn.Sequential`` of Linear / BatchNorm1d / ReLU / LeakyReLU / Dropout through the fused head kernels."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Linear):
            bn = act = None
            j = i + 1
            if j < len(mods) and isinstance(mods[j], nn.BatchNorm1d):
                bn = mods[j]; j += 1
            if j < len(mods) and isinstance(mods[j], (nn.ReLU, nn.LeakyReLU)):
                act = mods[j]; j += 1
            x = head_layer(x, m, bn, act)
            i = j
        else:
            x = m(x)                      # Dropout and anything else: the module itself
            i += 1
    return x
