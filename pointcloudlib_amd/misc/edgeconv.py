"""DGCNN's EdgeConv stage (kNN graph -> edge features -> Conv2d 1x1 + BatchNorm + LeakyReLU -> max over the neighbours;
/root/reference/networks/cls/dgcnn.py:29-50, :72-83, :100-111) without the [B,N,k,2C] edge tensor.

The conv is linear: with W = [Wa | Wb], ``W [x_nbr - x_i, x_i] = Wa x_nbr + (Wb - Wa) x_i = U[nbr] + V[i]`` where
``[U | V] = x [Wa ; Wb - Wa]^T`` is one plain GEMM over the N points (k = 20 times fewer flops than the conv over the
edges).  The rest -- BatchNorm batch statistics over all B*N*k edges, LeakyReLU, the max over the neighbours and their
backward -- runs in two streaming HIP kernels (csrc/edgeconv.hip).  Parameters live in a one-layer ``PointwiseMLP``
(weight [Cout, 2C], gamma, beta, running statistics), so the module tree and state dict are those of the dense path.
"""
import ctypes

import torch

from .. import _lib, syncbn
from .ops import _dev, _p, _stream, edge_features


class _EdgeConvPool(torch.autograd.Function):
    """UV [B,N,2C] (U | V), idx [B,N,k] -> max_j lrelu(BN(U[idx[.,j]] + V)) [B,N,C]."""

    @staticmethod
    def forward(ctx, UV, idx, gamma, beta, rmean, rvar, cfg, UVlo=None, cat_slice=None, want_t=False):
        """``cat_slice``: a [B,N,C] column slice (unit stride along C) of a wider [B,N,Ctot] buffer that also receives the output -- the
        concatenation of the stages' outputs is then written by the stages (``assemble``), not by a copy kernel."""
        # ``want_t``: also return the output transposed per cloud, [B,C,N] (non-differentiable; what the next stage's k-NN search reads)
        slope, eps, momentum, training = cfg
        UV = _dev(UV, "UV")
        idx = _dev(idx, "idx", torch.int32)
        B, N, C2 = UV.shape
        C, k = C2 // 2, idx.shape[2]
        dev = UV.device
        G = B * N
        st = _stream()
        rows = _lib.size_query("pcl_edgeconv_stat_rows", B, N)
        stats = torch.empty((rows, 2, C), dtype=torch.float64, device=dev)
        ymax, ymin = torch.empty((G, C), device=dev), torch.empty((G, C), device=dev)
        jmax, jmin = torch.empty((G, C), dtype=torch.int32, device=dev), torch.empty((G, C), dtype=torch.int32, device=dev)
        need_grad = ctx.needs_input_grad[0]
        sumU = torch.empty((G, C), device=dev) if need_grad else None          # SU[i] = sum_j U[nbr(i,j)] for the backward
        if UVlo is not None:
            _lib.call("pcl_edgeconv_gather_hilo_f32", _p(UV), _p(UVlo), _p(idx), B, N, k, C, _p(ymax), _p(ymin), _p(jmax), _p(jmin), _p(stats),
                      _p(sumU), st, algo_bytes=4 * G * (4 * C + k + 4 * C), tag=f"edge{C}")
        else:
            _lib.call("pcl_edgeconv_gather_f32", _p(UV), _p(idx), B, N, k, C, _p(ymax), _p(ymin), _p(jmax), _p(jmin), _p(stats),
                      _p(sumU), st, algo_bytes=4 * G * (2 * C + k + 4 * C), tag=f"edge{C}")
        in_off = in_src = None
        if need_grad and N <= 8192:
            # the neighbour lists transposed (who points at me), once per graph: the backward sums over them without atomics
            in_off = torch.empty((G + 1,), dtype=torch.int32, device=dev)
            in_src = torch.empty((G * k,), dtype=torch.int32, device=dev)
            _lib.call("pcl_knn_transpose_i32", _p(idx), B, N, k, _p(in_off), _p(in_src), st)
        if training:
            scale, shift, mean, invstd = (torch.empty((C,), device=dev) for _ in range(4))
            f_stats, f_rows, f_P = (stats, rows, G * k) if not syncbn.active() else syncbn.reduce_rows(stats, rows, G * k)
            _lib.call("pcl_bn_finalize_f32", _p(f_stats), f_rows, _p(gamma), _p(beta), f_P, C, eps, momentum, _p(scale), _p(shift),
                      _p(mean), _p(invstd), _p(rmean), _p(rvar), st)
        else:
            invstd = torch.rsqrt(rvar + eps)
            mean = rmean
            scale = gamma * invstd
            shift = beta - scale * rmean
        out = torch.empty((G, C), device=dev)
        arg = torch.empty((G, C), dtype=torch.int32, device=dev)
        ysel = torch.empty((G, C), device=dev)
        out_t = None
        if cat_slice is not None:
            if not (cat_slice.is_cuda and cat_slice.dtype == torch.float32 and tuple(cat_slice.shape) == (B, N, C) and cat_slice.stride(2) == 1
                    and cat_slice.stride(0) == N * cat_slice.stride(1)):
                raise ValueError("cat_slice: a [B,N,C] column slice of a contiguous [B,N,Ctot] float32 buffer")
        if want_t and N % 32 == 0:
            out_t = torch.empty((B, C, N), device=dev)
            _lib.call("pcl_group_minmax_finalize_t_f32", _p(ymax), _p(ymin), _p(jmax), _p(jmin), _p(scale), _p(shift), slope, B, N, C,
                      _p(out), _p(arg), _p(ysel), _p(cat_slice), 0 if cat_slice is None else cat_slice.stride(1), _p(out_t), st)
        elif cat_slice is not None:
            _lib.call("pcl_group_minmax_finalize2_f32", _p(ymax), _p(ymin), _p(jmax), _p(jmin), _p(scale), _p(shift), slope, G, C,
                      _p(out), _p(arg), _p(ysel), _p(cat_slice), cat_slice.stride(1), st)
        else:
            _lib.call("pcl_group_minmax_finalize_f32", _p(ymax), _p(ymin), _p(jmax), _p(jmin), _p(scale), _p(shift), slope, G, C,
                      _p(out), _p(arg), _p(ysel), st)
        ctx.cfg = (slope, training, B, N, k, C)
        ctx.save_for_backward(UV, idx, out, arg, ysel, gamma, mean, invstd, scale, in_off, in_src, sumU)
        if want_t:
            if out_t is None:
                out_t = out.view(B, N, C).transpose(1, 2).contiguous()
            ctx.mark_non_differentiable(out_t)
            ctx.set_materialize_grads(False)          # (else autograd fills a [B,C,N] zero tensor per stage for the transposed copy's gradient)
            return out.view(B, N, C), out_t
        return out.view(B, N, C)

    @staticmethod
    def backward(ctx, gout, *unused):
        slope, training, B, N, k, C = ctx.cfg
        UV, idx, out, arg, ysel, gamma, mean, invstd, scale, in_off, in_src, sumU = ctx.saved_tensors
        if gout is None:                              # (gradients are not materialised: nothing flowed into this stage's output)
            gout = torch.zeros((B, N, C), device=UV.device)
        dev = UV.device
        G = B * N
        st = _stream()
        gout = _dev(gout, "grad").reshape(G, C)
        gz = torch.empty((G, C), device=dev)
        stats = torch.empty((1024, 2, C), dtype=torch.float64, device=dev)
        rows_c = ctypes.c_int(0)
        _lib.call("pcl_maxgrad_prep_f32", _p(gout), _p(out), _p(ysel), slope, G, C, _p(gz), _p(stats), ctypes.byref(rows_c), st)
        rows = rows_c.value
        a, k1, k2 = (torch.empty((C,), device=dev) for _ in range(3))
        if training:
            dgamma, dbeta = torch.empty((C,), device=dev), torch.empty((C,), device=dev)
            _lib.call("pcl_bn_bwd_consts_f32", _p(stats), rows, _p(gamma), _p(mean), _p(invstd), G * k, C, _p(dgamma), _p(dbeta),
                      _p(a), _p(k1), _p(k2), None, st)
            if syncbn.active():
                # dgamma / dbeta stay sums over THIS rank's edges (the gradient all-reduce averages them); the constants of
                # dy = a*du - k1 - k2*(y - mean) come from the global sums over the global edge count (syncbn.py)
                g_stats, g_rows, g_P = syncbn.reduce_rows(stats, rows, G * k)
                _lib.call("pcl_bn_bwd_consts_f32", _p(g_stats), g_rows, _p(gamma), _p(mean), _p(invstd), g_P, C, None, None,
                          _p(a), _p(k1), _p(k2), None, st)
        else:
            s = stats[:rows].sum(0)
            a.copy_(scale); k1.zero_(); k2.zero_()
            dbeta = s[0].float()
            dgamma = ((s[1] - mean.double() * s[0]) * invstd.double()).float()
        dUV = torch.empty_like(UV)
        lists = in_off is not None
        _lib.call("pcl_edgeconv_scatter_f32", _p(UV), _p(idx), _p(gz), _p(arg), _p(a), _p(k1), _p(k2), _p(mean), B, N, k, C,
                  _p(in_off), _p(in_src), _p(sumU) if lists else None, _p(dUV), st)
        return dUV, None, dgamma, dbeta, None, None, None, None, None, None


class _Assembled(torch.autograd.Function):
    """``torch.cat(parts, dim=-1)`` whose result already sits in ``buf``: every part's producer wrote its column slice (``cat_slice``).
    Forward hands ``buf`` on, backward returns the column slices of the gradient -- what CatBackward returns."""

    @staticmethod
    def forward(ctx, buf, *parts):
        ctx.widths = [p.shape[-1] for p in parts]
        return buf.view(buf.shape)

    @staticmethod
    def backward(ctx, g):
        outs, o = [], 0
        for w in ctx.widths:
            outs.append(g[..., o:o + w])
            o += w
        return (None, *outs)


def assemble(buf, parts):
    """The concatenation of ``parts`` along the last axis, given that each part's producer has also written it into its column slice of
    ``buf`` (``edge_conv(..., cat_slice=buf[..., a:b])``)."""
    return _Assembled.apply(buf, *parts)


class _PointLinear(torch.autograd.Function):
    """UV [B,N,2*Cout] = x [B,N,C] @ [Wa ; Wb - Wa]^T for W = [Wa | Wb] [Cout, 2C], on the library's own GEMM kernels
    (forward, dX and dW); returns the gradient of W itself."""

    @staticmethod
    def forward(ctx, x, W, flush_k=0, hilo=False):
        x = _dev(x, "x")
        B, N, C = x.shape
        Co = W.shape[0]
        dev = x.device
        Wcat = torch.empty((2 * Co, C), device=dev)                                  # [Wa ; Wb - Wa]
        _lib.call("pcl_edgeconv_wcat_f32", _p(W.contiguous()), Co, C, 0, _p(Wcat), _stream())
        P = B * N
        UV = torch.empty((B, N, 2 * Co), device=dev)
        UVlo = None
        if flush_k:
            # y = U[nbr] + V decides the max-pool winners of the stage.  Chains of flush_k terms summed in fp64 give U | V as the fp32
            # rounding of the exact product and, with ``hilo``, its residual (UVlo): the gather then forms y from both (csrc/edgeconv.hip:
            # HILO) -- the fp32 rounding of the exact edge value; against fp64 on EQUAL inputs a stage is then 100 x closer than the edge
            # form in PyTorch-CPU fp32 (tools/dbg/edgeconv_err.py), for +5 % of the DGCNN step: opt-in
            if hilo:
                UVlo = torch.empty((B, N, 2 * Co), device=dev)
            _lib.call("pcl_frag_linear_fwd_f32", _p(x), C, _p(Wcat), C, None, None, None, 0.0, P, C, 2 * Co, _p(UV), 2 * Co, _p(UVlo), None, int(flush_k),
                      _stream(), tag=f"uv{C}x{2 * Co}")
        else:
            rows = _lib.size_query("pcl_mlp_stat_rows", P, 2 * Co, 0)
            _lib.call("pcl_linear_fwd_rows_f32", _p(x), _p(Wcat), None, None, None, 0.0, P, C, 2 * Co, _p(UV),
                      _p(torch.empty((rows, 2, 2 * Co), dtype=torch.float64, device=dev)), None, None, _stream(), tag=f"uv{C}x{2 * Co}")
        ctx.save_for_backward(x, Wcat)
        if UVlo is not None:
            ctx.mark_non_differentiable(UVlo)
            return UV, UVlo
        return UV

    @staticmethod
    def backward(ctx, dUV, *unused):
        x, Wcat = ctx.saved_tensors
        B, N, C = x.shape
        C2 = Wcat.shape[0]
        Co = C2 // 2
        dev = x.device
        P = B * N
        st = _stream()
        lib = _lib.lib()
        dUV = _dev(dUV, "grad")
        from .mlp_hip import _unit_consts
        one, zero = _unit_consts(dev, C2)                   # cached read-only constants: no fill launches per call
        dx = dW = None
        if ctx.needs_input_grad[1]:
            nbytes = _lib.size_query("pcl_linear_bwd_dw_workspace_bytes", P, C2, C)
            ws = torch.empty(((nbytes + 3) // 4,), device=dev)
            dWcat = torch.empty((C2, C), device=dev)
            _lib.call("pcl_linear_bwd_dw_rows_f32", _p(dUV), _p(dUV), _p(one), _p(zero), _p(zero), _p(zero), None, None, 1, _p(x),
                      None, None, 0.0, P, C2, C, _p(dWcat), _p(ws), nbytes, None, None, 0, st, tag=f"uvdw{C2}x{C}")
            dW = torch.empty((Co, 2 * C), device=dev)                              # Wcat = [Wa ; Wb - Wa]
            _lib.call("pcl_edgeconv_wcat_f32", _p(dWcat), Co, C, 1, _p(dW), st)
        if ctx.needs_input_grad[0]:
            dx = torch.empty((B, N, C), device=dev)
            _lib.call("pcl_linear_bwd_dx_rows_f32", _p(dUV), _p(dUV), _p(one), _p(zero), _p(zero), _p(zero), None, None, 1, _p(Wcat),
                      P, C2, C, None, None, None, 0.0, _p(dx), None, None, None, 0, 0, st, tag=f"uvdx{C2}x{C}")
        return dx, dW, None, None


def edge_conv(mlp, x, idx, cat_slice=None, want_t=False):
    """One EdgeConv stage on channel-last ``x`` [B,N,C] with neighbour lists ``idx`` [B,N,k] (int32) -> [B,N,Cout].
    ``mlp``: the stage's one-layer ``PointwiseMLP([2C, Cout], slope=0.2)``.  The HIP backend takes the factorised path;
    the plain-PyTorch backend (tests) and anything that is not a single bias-free conv+BN layer build the edge tensor."""
    k = idx.shape[2]
    if mlp.resolved_backend(x) == "hip" and mlp.n_layers == 1 and mlp.bn and mlp.biases is None and mlp.last_act:
        flush_k = int(getattr(mlp, "flush_k", 0))
        hilo = bool(flush_k) and bool(getattr(mlp, "edge_hilo", False))
        UV = _PointLinear.apply(x, mlp.weights[0], flush_k, hilo)             # one GEMM over the points
        UV, UVlo = UV if hilo else (UV, None)
        cfg = (mlp.slope, mlp.eps, mlp.momentum, mlp.training)
        return _EdgeConvPool.apply(UV, idx, mlp.gammas[0], mlp.betas[0], mlp.running_mean_0, mlp.running_var_0, cfg, UVlo, cat_slice, want_t)
    out = mlp(edge_features(x, idx), group_max=k)
    if cat_slice is not None:
        cat_slice.copy_(out.detach())
    return (out, out.detach().transpose(1, 2).contiguous()) if want_t else out


class _MaxMeanPoolBN(torch.autograd.Function):
    """Global max and mean pooling over the N points of each cloud with the producing conv's BatchNorm + LeakyReLU folded in
    (csrc/mlp.hip: bn_act_maxmean_sliced_kernel): ``Y`` [B,N,C] is the PRE-BatchNorm conv output of a deferring stack
    (mlp_hip.stack_plain_deferred), the result [B, 2C] = cat(max_n z, mean_n z), z = lrelu(scale*y + shift).  The backward hands
    the stack du and its BatchNorm-backward sums through ``link`` (the protocol of pointconv_utils._PointConvContractBN)."""

    @staticmethod
    def forward(ctx, Y, link):
        B, N, C = Y.shape
        out = torch.empty((B, 2 * C), dtype=torch.float32, device=Y.device)
        arg = torch.empty((B, C), dtype=torch.int32, device=Y.device)
        _lib.call("pcl_bn_act_max_mean_f32", _p(Y), _p(link.scale), _p(link.shift), float(link.slope), B, N, C, 2 * C, _p(out),
                  out.data_ptr() + 4 * C, _p(arg), _stream(), algo_bytes=4 * B * N * C)
        ctx.link = link
        ctx.save_for_backward(Y, arg, link.scale, link.shift)
        return out

    @staticmethod
    def backward(ctx, gout):
        import ctypes
        Y, arg, scale, shift = ctx.saved_tensors
        link = ctx.link
        B, N, C = Y.shape
        gout = _dev(gout, "grad")
        du = torch.empty_like(Y)
        stats = torch.empty((1024, 2, C), dtype=torch.float64, device=Y.device)
        rows = ctypes.c_int(0)
        _lib.call("pcl_bn_act_max_mean_bwd_f32", _p(gout), gout.data_ptr() + 4 * C, 2 * C, _p(arg), _p(Y), _p(scale), _p(shift),
                  float(link.slope), B, N, C, _p(du), _p(stats), ctypes.byref(rows), _stream(), algo_bytes=8 * B * N * C)
        link.stats, link.rows = stats, rows.value
        return du, None


def conv_max_mean_pool(mlp, x):
    """``y = mlp(x)`` (one conv + BatchNorm + LeakyReLU on [B,N,Cin]) followed by ``cat(y.max(1), y.mean(1))`` -> [B, 2*Cout]
    (networks/cls/dgcnn.py:113-116).  On the HIP training path the activated [B,N,Cout] tensor is never formed: the conv's stack
    stops at its pre-BatchNorm output and the pooling kernel applies BatchNorm + activation while it reduces."""
    from . import mlp_hip
    B, N, Cin = x.shape
    if x.is_cuda and x.dtype == torch.float32 and mlp.resolved_backend(x) == "hip" and mlp.last_act and B <= 65535:      # (the clouds are the pooling kernels' grid.y)
        r = mlp_hip.stack_plain_deferred(mlp, x.reshape(B * N, Cin).contiguous())
        if r is not None:
            Y, link = r
            return _MaxMeanPoolBN.apply(Y.view(B, N, Y.shape[-1]), link)
    y = mlp(x)
    return torch.cat((y.max(dim=1)[0], y.mean(dim=1)), dim=1)
