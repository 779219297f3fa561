"""The FC classification head on a handful of rows (one per cloud) -- host side of csrc/head.hip.

The counterpart networks keep the reference's module tree (``nn.Linear`` / ``nn.BatchNorm1d`` / ``nn.ReLU`` /
``nn.Dropout``; networks/cls/pointnet2.py:138-147, dgcnn.py:87-93, pointnet.py:22-38) as parameter containers; on the GPU
``head_layer`` / ``fc_head`` run Linear + BatchNorm1d + activation as one kernel per layer (forward) and two (backward)
instead of ~15 library launches per layer pair.  More than 64 rows fall back to the PyTorch modules (plain library GEMMs).
"""
import ctypes
import os

import torch
import torch.nn.functional as F
from torch import nn

from .. import _lib, syncbn
from .ops import _p, _stream

MAX_ROWS = 64
USE_STACK = os.environ.get("PCL_STACK", "1") != "0"       # the whole head as one C call per direction (pcl_fc_head_*_f32)
_MAXL = 4


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
        nws = _lib.size_query("pcl_head_layer_fwd_workspace_bytes", R, K, N)      # split-K partial sums of a wide layer (else 0)
        ws = torch.empty((nws // 4,), device=dev) if nws else None
        _lib.call("pcl_head_layer_fwd_f32", _p(x), _p(W), _p(b), _p(gamma), _p(beta), _p(rmean), _p(rvar), R, K, N, bn_mode, eps,
                  momentum, slope, _p(ypre), _p(out), _p(mean), _p(invstd), _p(ws), nws, _stream())
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
    def synced(y):
        # dp.FlatBucketDP(sync_bn=True) checks that every training-mode BatchNorm module came through HERE: the mark is set only where
        # the synchronised statistics are actually used (it used to be set before the per-rank ``bn(y)`` branch below)
        bn._pcl_sync_routed = True
        return syncbn.batch_norm_1d(y, bn)
    if x.shape[0] > MAX_ROWS or x.dim() != 2 or x.dtype != torch.float32:     # large batches: plain library GEMMs
        y = linear(x)
        if bn is not None:
            y = synced(y) if (bn.training and syncbn.active() and y.dim() == 2) else bn(y)
        return y if slope == 1.0 else F.leaky_relu(y, slope)
    if bn is not None and bn.training and syncbn.active():
        # synchronised statistics over all ranks' rows: the one-kernel layer computes its batch statistics inside the kernel,
        # so this case (R <= 64 rows per rank) runs the library GEMM + syncbn.batch_norm_1d (fp64 sums, biased running variance)
        y = synced(linear(x))
        return y if slope == 1.0 else F.leaky_relu(y, slope)
    if bn is None:
        cfg = (0, 0.0, 0.0, slope)
        return _HeadLayer.apply(x, linear.weight, linear.bias, None, None, None, None, cfg)
    # | 4: running_var follows Jittor's rule (biased batch variance) like PointwiseMLP / pcl_bn_finalize_f32
    mode = (1 if bn.training else 2) | 4
    momentum = 0.1 if bn.momentum is None else bn.momentum
    cfg = (mode, bn.eps, momentum, slope)
    return _HeadLayer.apply(x, linear.weight, linear.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, cfg)


class _CHeadLayer(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_void_p) for n in ("W", "bias", "gamma", "beta", "running_mean", "running_var", "dW", "dbias", "dgamma", "dbeta")]
                + [(n, ctypes.c_int32) for n in ("K", "N", "bn_mode")] + [(n, ctypes.c_float) for n in ("eps", "momentum", "slope", "drop_p")]
                + [("pad_", ctypes.c_int32)])


class _CHead(ctypes.Structure):
    _fields_ = [("struct_bytes", ctypes.c_int32), ("n_layers", ctypes.c_int32), ("R", ctypes.c_int32), ("pad_", ctypes.c_int32),
                ("seed", ctypes.c_uint64), ("x", ctypes.c_void_p), ("layer", _CHeadLayer * _MAXL), ("out", ctypes.c_void_p),
                ("save", ctypes.c_void_p), ("save_bytes", ctypes.c_size_t), ("tmp", ctypes.c_void_p), ("tmp_bytes", ctypes.c_size_t),
                ("gout", ctypes.c_void_p), ("dx", ctypes.c_void_p), ("stream", ctypes.c_void_p)]


class _HeadPlan:
    __slots__ = ("desc", "ref", "save_bytes", "bwd_tmp", "layers", "gsizes", "gtotal", "R", "n_out", "params_of")


_HEAD_PLANS = {}
_DROP_CALLS = [0]


def _head_plan(layers, R, training):
    """layers: [(linear, bn | None, slope, drop_p)]"""
    key = (tuple(id(l[0]) for l in layers), R, training,
           tuple((l[2], l[3], l[0].bias is not None, None if l[1] is None else (l[1].eps, l[1].momentum, l[1].weight is not None)) for l in layers))
    plan = _HEAD_PLANS.get(key)
    if plan is not None and all(a[0] is b[0] and a[1] is b[1] for a, b in zip(plan.layers, layers)):
        return plan
    d = _CHead()
    d.struct_bytes, d.n_layers, d.R = ctypes.sizeof(_CHead), len(layers), R
    sizes = []
    for i, (lin, bn, slope, drop_p) in enumerate(layers):
        ly = d.layer[i]
        ly.W = 1
        ly.K, ly.N = lin.in_features, lin.out_features
        if bn is None:
            ly.bn_mode, ly.eps, ly.momentum = 0, 0.0, 0.0
        else:
            # | 4: running_var follows Jittor's rule (biased batch variance) like PointwiseMLP / pcl_bn_finalize_f32
            ly.bn_mode, ly.eps, ly.momentum = (1 if training else 2) | 4, bn.eps, (0.1 if bn.momentum is None else bn.momentum)
        ly.slope, ly.drop_p = slope, (drop_p if training else 0.0)
        sizes.append(lin.out_features * lin.in_features)
        if lin.bias is not None:
            sizes.append(lin.out_features)
        if bn is not None and bn.weight is not None:
            sizes += [lin.out_features, lin.out_features]
    sv, bt = ctypes.c_size_t(), ctypes.c_size_t()
    _lib.check(_lib.lib().pcl_fc_head_sizes(ctypes.byref(d), ctypes.byref(sv), ctypes.byref(bt)), "pcl_fc_head_sizes")
    plan = _HeadPlan()
    plan.desc, plan.ref, plan.save_bytes, plan.bwd_tmp = d, ctypes.byref(d), sv.value, bt.value
    plan.layers, plan.gsizes, plan.gtotal, plan.R, plan.n_out = layers, sizes, sum(sizes), R, layers[-1][0].out_features
    _HEAD_PLANS[key] = plan
    return plan


class _HeadStack(torch.autograd.Function):
    """x [R, K0] -> [R, N_last]; params per layer: W (, bias) (, gamma, beta) in that order; running statistics through aux."""

    @staticmethod
    def forward(ctx, x, aux, *params):
        plan, seed = aux
        d = type(plan.desc).from_buffer_copy(plan.desc)      # private copy per call (see mlp_hip._stack_plan's callers)
        x = x.contiguous()
        dev = x.device
        save = torch.empty((plan.save_bytes,), dtype=torch.uint8, device=dev)
        out = torch.empty((plan.R, plan.n_out), dtype=torch.float32, device=dev)
        i = 0
        for l, (lin, bn, _, _) in enumerate(plan.layers):
            ly = d.layer[l]
            ly.W = params[i].data_ptr(); i += 1
            if lin.bias is not None:
                ly.bias = params[i].data_ptr(); i += 1
            else:
                ly.bias = None
            if bn is not None:
                if bn.weight is not None:
                    ly.gamma, ly.beta = params[i].data_ptr(), params[i + 1].data_ptr(); i += 2
                else:
                    ly.gamma = ly.beta = None
                ly.running_mean = None if bn.running_mean is None else bn.running_mean.data_ptr()
                ly.running_var = None if bn.running_var is None else bn.running_var.data_ptr()
        d.seed, d.x, d.out, d.save, d.save_bytes, d.stream = seed, x.data_ptr(), out.data_ptr(), save.data_ptr(), plan.save_bytes, _stream()
        _lib.call("pcl_fc_head_fwd_f32", ctypes.byref(d), tag="head_fwd")
        ctx.plan, ctx.seed = plan, seed
        ctx.save_for_backward(x, out, save, *params)
        return out

    @staticmethod
    def backward(ctx, gout):
        plan = ctx.plan
        d = type(plan.desc).from_buffer_copy(plan.desc)      # private copy per call (see mlp_hip._stack_plan's callers)
        sv = ctx.saved_tensors
        x, out, save, params = sv[0], sv[1], sv[2], sv[3:]
        dev = x.device
        gout = gout.contiguous()
        tmp = torch.empty((plan.bwd_tmp,), dtype=torch.uint8, device=dev)
        flat = torch.empty((plan.gtotal,), dtype=torch.float32, device=dev)
        pieces = flat.split_with_sizes(plan.gsizes)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        base, o, i = flat.data_ptr(), 0, 0
        grads = []
        for l, (lin, bn, _, _) in enumerate(plan.layers):
            ly = d.layer[l]
            N, K = lin.out_features, lin.in_features
            ly.W = params[i].data_ptr()
            ly.dW = base + 4 * o; o += N * K
            grads.append(pieces[i].view(N, K)); i += 1
            if lin.bias is not None:
                ly.bias = params[i].data_ptr()
                ly.dbias = base + 4 * o; o += N
                grads.append(pieces[i]); i += 1
            else:
                ly.bias = ly.dbias = None
            if bn is not None and bn.weight is not None:
                ly.gamma, ly.beta = params[i].data_ptr(), params[i + 1].data_ptr()
                ly.dgamma = base + 4 * o; o += N
                ly.dbeta = base + 4 * o; o += N
                grads += [pieces[i], pieces[i + 1]]; i += 2
            else:
                ly.dgamma = ly.dbeta = None
        d.seed, d.x, d.out, d.save, d.save_bytes = ctx.seed, x.data_ptr(), out.data_ptr(), save.data_ptr(), plan.save_bytes
        d.tmp, d.tmp_bytes, d.gout, d.dx, d.stream = tmp.data_ptr(), plan.bwd_tmp, gout.data_ptr(), (None if dx is None else dx.data_ptr()), _stream()
        _lib.call("pcl_fc_head_bwd_f32", ctypes.byref(d), tag="head_bwd")
        return (dx, None) + tuple(grads)


def _head_layers(mods):
    """[Linear, BatchNorm1d?, (Leaky)ReLU?, Dropout?]* -> [(linear, bn, slope, drop_p)] or None when the sequence has anything else"""
    out, i = [], 0
    while i < len(mods):
        m = mods[i]
        if not isinstance(m, nn.Linear):
            return None
        bn, slope, p = None, 1.0, 0.0
        j = i + 1
        if j < len(mods) and isinstance(mods[j], nn.BatchNorm1d):
            bn = mods[j]; j += 1
        if j < len(mods) and isinstance(mods[j], (nn.ReLU, nn.LeakyReLU)):
            slope = _slope_of(mods[j]); j += 1
        if j < len(mods) and isinstance(mods[j], nn.Dropout):
            p = float(mods[j].p); j += 1
        out.append((m, bn, slope, p))
        i = j
    return out


def _dropout_seed(device, n_draws):
    """64-bit seed of one call's dropout masks (the kernels hash (seed, layer, element)), taken from torch's CUDA generator of the
    device the way ``nn.Dropout`` takes its Philox (seed, offset) pair: ``torch.cuda.manual_seed`` / ``fork_rng`` reset and restore
    the sequence, every call advances the generator's offset, and the rank of a data-parallel job is mixed in so that ranks that
    were seeded alike draw different masks for their different clouds."""
    rank = 0
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            rank = dist.get_rank()
    except Exception:
        pass
    try:
        gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
        base, off = gen.initial_seed(), gen.get_offset()
        gen.set_offset(off + 4 * ((int(n_draws) + 3) // 4))
    except Exception:                                # (a build without generator offsets: a process-wide call counter)
        _DROP_CALLS[0] += 1
        base, off = torch.initial_seed(), _DROP_CALLS[0]
    m = 0xFFFFFFFFFFFFFFFF
    z = (base * 0x9E3779B97F4A7C15 + off * 0xD1B54A32D192ED03 + (rank + 1) * 0x94D049BB133111EB) & m
    z ^= z >> 31
    return (z * 0xBF58476D1CE4E5B9) & m


def head_stack(mods, x):
    """The module sequence ``mods`` (Linear / BatchNorm1d / ReLU / LeakyReLU / Dropout) on x [R, K] as ONE autograd node and one
    C call per direction, or None when this path does not apply (then the per-layer path runs)."""
    if not (USE_STACK and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] <= MAX_ROWS):
        return None
    layers = _head_layers(mods)
    if not layers or len(layers) > _MAXL:
        return None
    training = mods[0].training
    if any(bn is not None and bn.training != training for _, bn, _, _ in layers):
        return None
    if training and syncbn.active() and any(bn is not None for _, bn, _, _ in layers):
        return None                                   # synchronised statistics: the per-layer path (syncbn.batch_norm_1d)
    if any(p >= 1.0 for _, _, _, p in layers):
        return None
    plan = _head_plan(layers, x.shape[0], training)
    params = []
    for lin, bn, _, _ in layers:
        params.append(lin.weight)
        if lin.bias is not None:
            params.append(lin.bias)
        if bn is not None and bn.weight is not None:
            params += [bn.weight, bn.bias]
    seed = 0
    if training and any(p > 0.0 for _, _, _, p in layers):
        seed = _dropout_seed(x.device, sum(l[0].out_features for l in layers) * x.shape[0])
    return _HeadStack.apply(x, (plan, seed), *params)


def fc_head(seq, x):
    """Run an ``nn.Sequential`` of Linear / BatchNorm1d / ReLU / LeakyReLU / Dropout through the fused head kernels."""
    mods = list(seq)
    y = head_stack(mods, x)
    if y is not None:
        return y
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
