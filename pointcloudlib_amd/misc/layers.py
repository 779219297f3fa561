"""Per-group pointwise MLP blocks -- host-side mirror of the 1x1-conv + BatchNorm + activation stacks
of the reference (``build_mlps`` networks/cls/pointnet2.py:18-31; ``Dense_Conv1d/2d``, ``Conv``
misc/layers.py:173-270; DGCNN conv1-4 networks/cls/dgcnn.py:72-83).

The reference transposes to NCHW and calls ``nn.Conv(kernel_size=1)``; a 1x1 conv is a row-wise linear
map, so here activations stay channel-last ``[..., C]`` and each layer is ``Y = X W^T (+b)`` over
P = prod(leading dims) rows, followed by training-mode BatchNorm over all P rows and ReLU/LeakyReLU.

BatchNorm semantics (Jittor's ``nn.BatchNorm``): batch mean and *biased*
variance ``max(E[x^2]-E[x]^2, 0)``, eps 1e-5, running stats ``r += (batch - r) * momentum`` with
momentum 0.1 and the biased variance; affine gamma=1, beta=0 at init.

One implementation ships: ``backend="hip"`` (= ``"auto"``, the default) -- the fused gfx950 kernels of libpcl_hip.so; CPU
tensors raise, there is no fallback.  Tests compare it with a plain-PyTorch composite of the same layers; that composite is
test infrastructure and lives in ``oracle/torch_backend.py``, which registers itself here under the name ``"torch"``
(``register_reference_backend``).  Nothing in this package imports it, and a module whose ``backend`` names an implementation
that nobody registered raises.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

__all__ = ["PointwiseMLP", "batch_norm_train", "max_over_group", "register_reference_backend", "reference_backend"]

# name -> object with ``mlp_forward(mlp, x, group_max)``, ``batch_norm_rows(x2d, gamma, beta, running_mean, running_var,
# momentum, eps)`` and ``sepconv_forward_x(module, X, F1, F2)``.  Empty in the product; filled by test infrastructure.
_REFERENCE_BACKENDS = {}


def register_reference_backend(name, impl):
    if name in ("hip", "auto"):
        raise ValueError(f"{name!r} names the library's own kernels")
    _REFERENCE_BACKENDS[name] = impl


def reference_backend(name):
    try:
        return _REFERENCE_BACKENDS[name]
    except KeyError:
        raise RuntimeError(f"backend {name!r} is not part of pointcloudlib_amd: the library runs its HIP kernels only (no fallback).  The "
                           "plain-PyTorch composite the tests compare against is registered by `import oracle.torch_backend`.") from None


class _BNRows(torch.autograd.Function):
    """Training-mode BatchNorm over the rows of x [P,C] on the library's kernels (csrc/mlp.hip: bn_rows_*): 3 launches forward,
    3 backward -- fp64 column sums, the stacks' finalize / constants kernels (biased running variance, Jittor's rule), one apply
    pass each way.  The plain-PyTorch composite below needed ~11 launches per BatchNorm."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps):
        import ctypes
        from .. import _lib
        from .ops import _p, _stream
        x = x.contiguous()
        P, C = x.shape
        dev = x.device
        st = _stream()
        stats = torch.empty((1024, 2, C), dtype=torch.float64, device=dev)
        rows = ctypes.c_int(0)
        _lib.call("pcl_bn_rows_stats_f32", _p(x), None, P, C, _p(stats), ctypes.byref(rows), st)
        vec = torch.empty((4, C), dtype=torch.float32, device=dev)
        scale, shift, mean, invstd = vec.unbind(0)
        _lib.call("pcl_bn_finalize_f32", _p(stats), rows.value, _p(gamma), _p(beta), P, C, eps, momentum, _p(scale), _p(shift), _p(mean),
                  _p(invstd), _p(running_mean), _p(running_var), st)
        out = torch.empty_like(x)
        _lib.call("pcl_bn_act_f32", _p(x), _p(scale), _p(shift), 1.0, P, C, _p(out), st)
        ctx.save_for_backward(x, gamma, mean, invstd)
        return out

    @staticmethod
    def backward(ctx, g):
        import ctypes
        from .. import _lib
        from .ops import _p, _stream
        x, gamma, mean, invstd = ctx.saved_tensors
        P, C = x.shape
        dev = x.device
        st = _stream()
        g = g.contiguous()
        stats = torch.empty((1024, 2, C), dtype=torch.float64, device=dev)
        rows = ctypes.c_int(0)
        _lib.call("pcl_bn_rows_stats_f32", _p(x), _p(g), P, C, _p(stats), ctypes.byref(rows), st)
        vec = torch.empty((5, C), dtype=torch.float32, device=dev)
        a, k1, k2, dgamma, dbeta = vec.unbind(0)
        _lib.call("pcl_bn_bwd_consts_f32", _p(stats), rows.value, _p(gamma), _p(mean), _p(invstd), P, C, _p(dgamma), _p(dbeta), _p(a), _p(k1),
                  _p(k2), None, st)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.call("pcl_bn_rows_bwd_apply_f32", _p(g), _p(x), _p(a), _p(k1), _p(k2), _p(mean), P, C, _p(dx), st)
        return dx, dgamma, dbeta, None, None, None, None


def batch_norm_train(x2d, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5):
    """BatchNorm over the rows of ``x2d`` [P,C] with Jittor's running-stat rule (biased variance) on the library's kernels
    (``_BNRows``).  Tensors the kernels do not take (CPU, fp64: only tests have them) go to the registered reference composite."""
    if not training:
        return F.batch_norm(x2d, running_mean, running_var, gamma, beta, False, 0.0, eps)
    from .. import syncbn
    if syncbn.active():          # data-parallel step with synchronised statistics: sums over every rank's rows
        return syncbn.batch_norm_rows(x2d, gamma, beta, running_mean, running_var, momentum, eps)
    if x2d.is_cuda and x2d.dtype == torch.float32 and gamma is not None and running_mean is not None:
        return _BNRows.apply(x2d, gamma, beta, running_mean, running_var, float(momentum), float(eps))
    return reference_backend("torch").batch_norm_rows(x2d, gamma, beta, running_mean, running_var, momentum, eps)


def max_over_group(x, dim):
    """``Var.argmax(dim)[1]`` (networks/cls/pointnet2.py:57) / ``x.max(dim=-1)`` (dgcnn.py:102): max VALUE."""
    return x.max(dim=dim)[0]


def set_accumulation(root, flush_k):
    """Select the accumulation of every ``PointwiseMLP`` under ``root`` (see ``PointwiseMLP.flush_k``): 0, 8 or 32."""
    if flush_k not in (0, 8, 32):
        raise ValueError(f"flush_k = {flush_k}: 0 (fp32 chains), 8 or 32 (fp64 sum of fp32 chains of that length)")
    for m in root.modules():
        if isinstance(m, PointwiseMLP):
            m.flush_k = int(flush_k)
    return root


class PointwiseMLP(nn.Module):
    """Stack of [Linear(Cin->Cout, bias) -> BatchNorm(train) -> (Leaky)ReLU] on channel-last rows.

    ``spec`` = [C0, C1, ..., CL].  ``bias=False`` mirrors ``nn.Conv(..., bias=not bn)``
    (networks/cls/pointnet2.py:26); ``slope`` 0 = ReLU, 0.2 = DGCNN's LeakyReLU."""

    def __init__(self, spec, bias=False, bn=True, slope=0.0, momentum=0.1, eps=1e-5, backend="auto",
                 last_act=True):
        super().__init__()
        self.spec = list(spec)
        self.bn = bn
        self.slope = float(slope)
        self.momentum = momentum
        self.eps = eps
        self.backend = backend
        self.last_act = last_act
        # accumulation of the forward GEMMs that run on plain rows: 0 = one fp32 fma chain per output (the staged MFMA kernels);
        # 8 | 32 = chains of at most that many terms summed in fp64 (csrc/frag.hip) -- set with ``set_accumulation`` by the networks
        # whose distance from the fp64 evaluation is accumulation error (part-seg decoders, DGCNN's stage products)
        self.flush_k = 0
        self.weights = nn.ParameterList()
        self.biases = nn.ParameterList() if bias else None
        self.gammas = nn.ParameterList()
        self.betas = nn.ParameterList()
        for i in range(1, len(spec)):
            cin, cout = spec[i - 1], spec[i]
            w = torch.empty(cout, cin)
            nn.init.kaiming_uniform_(w, a=math.sqrt(5))          # torch's Conv/Linear default
            self.weights.append(nn.Parameter(w))
            if bias:
                bound = 1.0 / math.sqrt(cin)
                self.biases.append(nn.Parameter(torch.empty(cout).uniform_(-bound, bound)))
            if bn:
                self.gammas.append(nn.Parameter(torch.ones(cout)))
                self.betas.append(nn.Parameter(torch.zeros(cout)))
                self.register_buffer(f"running_mean_{i - 1}", torch.zeros(cout))
                self.register_buffer(f"running_var_{i - 1}", torch.ones(cout))

    @property
    def n_layers(self):
        return len(self.weights)

    def resolved_backend(self, x):
        # "auto" IS the HIP path: a CPU tensor is an error there (no silent fallback).  Any other name must have been registered
        # by test infrastructure (register_reference_backend).
        return "hip" if self.backend == "auto" else self.backend

    def forward_grouped(self, xyz, new_xyz, feature, idx, cnt, group_off, use_xyz=True):
        """Ball-query grouping + this MLP + max over each group -> [B, m, C_last] (HIP backend): duplicate-compacted rows,
        and the first conv folded into the grouping where possible (mlp_hip.grouped_mlp); otherwise the rows are built
        by ``ops.group_points_compact`` and run through ``forward``."""
        from . import mlp_hip
        from .ops import group_points_compact
        if mlp_hip.can_fold_first_layer(self, use_xyz, feature):
            return mlp_hip.grouped_mlp(self, xyz, new_xyz, feature, idx, cnt, group_off, use_xyz)
        rows, rowset = group_points_compact(xyz, new_xyz, feature, idx, cnt, use_xyz, group_off=group_off)
        return self.forward(rows, rowset=rowset, x_grad_from=3 if use_xyz else 0)

    def forward(self, x, group_max=None, rowset=None, x_grad_from=0):
        """x [..., C0] -> [..., CL]; with ``group_max=ns`` the rows are groups of ns consecutive rows and
        the result is max-reduced over each group ([B,m,ns,C] -> [B,m,C]).  ``rowset``: duplicate-compacted
        ball-query groups (ops.group_points_compact), HIP backend only."""
        backend = self.resolved_backend(x)
        if rowset is not None and backend != "hip":
            raise RuntimeError("duplicate-compacted rows are a HIP-backend feature")
        if backend == "hip":
            K = x.shape[-1]
            from .. import syncbn
            if (group_max is None and rowset is None and self.n_layers == 1 and self.bn and K >= 2048 and K % 4 == 0
                    and x.numel() // K <= 32 and x.is_cuda and not (self.training and syncbn.active())):
                # (synchronised BatchNorm: the wide head kernel takes its batch statistics inside the kernel, so that case
                #  runs the fused MLP path below, whose statistics go through syncbn.reduce_rows)
                # a handful of rows against a long reduction (PointConv's Linear(16*C, C) on the GroupAll level: 32 x 16384):
                # 128x64 GEMM tiles give 16 workgroups here; the head kernels (csrc/head.hip) share X between 8 columns per
                # workgroup and stream the weight once.  +4: running_var takes the biased variance like pcl_bn_finalize_f32.
                from .head import _HeadLayer
                cfg = ((1 if self.training else 2) | 4, self.eps, self.momentum, self.slope if self.last_act else 1.0)
                y = _HeadLayer.apply(x.reshape(-1, K), self.weights[0], None if self.biases is None else self.biases[0],
                                     self.gammas[0], self.betas[0], self.running_mean_0, self.running_var_0, cfg)
                return y.reshape(*x.shape[:-1], y.shape[-1])
            from . import mlp_hip
            return mlp_hip.pointwise_mlp(self, x, group_max, rowset, x_grad_from)
        return reference_backend(backend).mlp_forward(self, x, group_max)
