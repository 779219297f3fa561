"""Fused gfx950 path of ``PointwiseMLP``: one autograd node for the whole [Linear -> BatchNorm -> (Leaky)ReLU] x L
(+ max over the group) stack, forward and backward, on the MFMA kernels of ``csrc/mlp.hip``.

Per layer the only P-sized tensors that touch HBM are the pre-BatchNorm outputs ``y_l`` (kept for backward) and,
in backward, the masked gradients ``du_l``; BatchNorm + activation are folded into operand staging, batch
statistics and BatchNorm-backward sums come out of GEMM epilogues (see the header of mlp.hip).
"""
import ctypes
import os

import torch

from .. import _lib, syncbn

_P = lambda t: None if t is None else t.data_ptr()  # noqa: E731
_FUSED_BWD = os.environ.get("PCL_FUSED_BWD", "1") != "0"      # lab switch: 0 = separate dX and dW kernels


def _stream():
    # the raw hipStream_t of torch's current stream (torch.cuda.current_stream().cuda_stream costs ~10 us of Python per call)
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def _rows_cost(nrows_dev, P, per_row, const):
    """Algorithmic bytes/flops of a row-streaming kernel: per_row * rows + const.  With duplicate-compacted rows the
    row count lives on the device; return a thunk the profiler resolves after the timed region."""
    if nrows_dev is None:
        return per_row * P + const
    return lambda: per_row * int(nrows_dev.item()) + const


def _empty(shape, dev, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=dev)


class _Pack:
    """Flatten named tensors / lists of tensors (None allowed) for ctx.save_for_backward and rebuild them."""

    def __init__(self):
        self.tensors, self.layout = [], []

    def add(self, name, t):
        self.layout.append((name, None if t is None else len(self.tensors), False))
        if t is not None:
            self.tensors.append(t)

    def add_list(self, name, ts):
        idx = []
        for t in ts:
            idx.append(None if t is None else len(self.tensors))
            if t is not None:
                self.tensors.append(t)
        self.layout.append((name, idx, True))

    @staticmethod
    def unpack(layout, saved):
        out = {}
        for name, idx, is_list in layout:
            if is_list:
                out[name] = [None if i is None else saved[i] for i in idx]
            else:
                out[name] = None if idx is None else saved[idx]
        return out


class _FusedMLP(torch.autograd.Function):
    """inputs: x [P,C0]; cfg = (group_ns|0, slope, eps, momentum, training, bn, last_act); then per layer
    (W, bias|None, gamma|None, beta|None, running_mean|None, running_var|None)."""

    @staticmethod
    def forward(ctx, x, cfg, *params):
        ns, slope, eps, momentum, training, bn, last_act, rowset, x_grad_from, link = cfg
        L = len(params) // 6
        dev = x.device
        x = x.contiguous()
        P, C0 = x.shape                      # P = row capacity; with a RowSet only the first *n_rows_dev rows are valid
        Pbn = P                              # rows BatchNorm averages over (padded duplicates count, see RowSet)
        rmeta = nrows = None
        if rowset is not None:
            assert ns == rowset.ns and P == rowset.capacity
            rmeta, nrows = rowset.row_meta, rowset.n_rows_dev
        st = _stream()
        Ys, scales, shifts, means, invstds = [], [], [], [], []
        cur, in_scale, in_shift = x, None, None
        cin = C0
        W0p = None
        for l in range(L):
            W, bias, gamma, beta, rmean, rvar = params[6 * l:6 * l + 6]
            cout = W.shape[0]
            if l == 0 and link is None and W.shape[1] != cin:
                # the input rows carry zero columns beyond the layer's fan-in (ops.group_points_compact pads rows to a
                # multiple of 4 floats so that every kernel moves 16-byte pieces): pad W with zero columns to match
                assert W.shape[1] < cin, f"input has {cin} columns, first layer expects {W.shape[1]}"
                W = W0p = torch.nn.functional.pad(W.detach(), (0, cin - W.shape[1]))
            fused_max = (l == L - 1) and ns in (32, 64) and rowset is None
            folded = l == 0 and link is not None
            if folded:
                # the first layer was folded into the grouping (_GroupLinear): x IS its pre-BatchNorm output
                assert cout == cin and L >= 2, "folded first layer: x must be the layer's own output"
                Y, stats, rows = x, link.stats, link.rows
            else:
                Y = _empty((P, cout), dev)
                rows = _lib.size_query("pcl_mlp_stat_rows", P, cout, 2 if rowset is not None else 0)
                stats = _empty((rows, 2, cout), dev, torch.float64)
            if folded:
                pass                       # nothing to launch: the grouping kernel already produced Y and its sums
            elif fused_max:      # last layer of a max-pooled stack: per-group min/max come out of the GEMM epilogue
                G = P // ns
                gmax, gmin = _empty((G, cout), dev), _empty((G, cout), dev)
                gamax, gamin = _empty((G, cout), dev, torch.int32), _empty((G, cout), dev, torch.int32)
                _lib.call("pcl_linear_fwd_gmax_f32", _P(cur), _P(W), _P(bias), _P(in_scale), _P(in_shift), slope, P, cin,
                          cout, ns, _P(Y), _P(stats), _P(gmax), _P(gmin), _P(gamax), _P(gamin), st,
                          algo_bytes=4 * P * (cin + cout) + 4 * cin * cout, algo_flops=2 * P * cin * cout,
                          tag=f"fwd{cin}x{cout}")
            else:
                _lib.call("pcl_linear_fwd_rows_f32", _P(cur), _P(W), _P(bias), _P(in_scale), _P(in_shift), slope, P, cin, cout,
                          _P(Y), _P(stats), _P(rmeta), _P(nrows), st,
                          algo_bytes=_rows_cost(nrows, P, 4 * (cin + cout), 4 * cin * cout),
                          algo_flops=_rows_cost(nrows, P, 2 * cin * cout, 0), tag=f"fwd{cin}x{cout}")
            if bn and training:
                scale, shift, mean, invstd = _empty((4, cout), dev).unbind(0)     # one allocation: the host enqueues ~2 us per torch.empty
                f_stats, f_rows, f_P = (stats, rows, Pbn) if not syncbn.active() else syncbn.reduce_rows(stats, rows, Pbn)
                _lib.call("pcl_bn_finalize_f32", _P(f_stats), f_rows, _P(gamma), _P(beta), f_P, cout, eps, momentum, _P(scale),
                          _P(shift), _P(mean), _P(invstd), _P(rmean), _P(rvar), st)
            elif bn:
                invstd = torch.rsqrt(rvar + eps)
                mean = rmean
                scale = gamma * invstd
                shift = beta - scale * rmean
            else:
                scale, shift = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
                mean, invstd = shift, scale
            Ys.append(Y); scales.append(scale); shifts.append(shift); means.append(mean); invstds.append(invstd)
            cur, in_scale, in_shift, cin = Y, scale, shift, cout
        out_slope = slope if last_act else 1.0
        if ns:
            G = P // ns
            out = _empty((G, cin), dev)
            arg = _empty((G, cin), dev, torch.int32)
            ymax = _empty((G, cin), dev)
            if fused_max:
                _lib.call("pcl_group_minmax_finalize_f32", _P(gmax), _P(gmin), _P(gamax), _P(gamin), _P(in_scale),
                          _P(in_shift), out_slope, G, cin, _P(out), _P(arg), _P(ymax), st)
            elif rowset is not None:
                _lib.call("pcl_bn_act_max_rows_f32", _P(cur), _P(rowset.group_off), _P(in_scale), _P(in_shift), out_slope, G,
                          cin, _P(out), _P(arg), _P(ymax), st, algo_bytes=_rows_cost(nrows, P, 4 * cin, 12 * G * cin),
                          tag=f"maxrows{cin}")
            else:
                _lib.call("pcl_bn_act_max_f32", _P(cur), _P(in_scale), _P(in_shift), out_slope, G, ns, cin, _P(out), _P(arg),
                          _P(ymax), st, algo_bytes=4 * P * cin + 12 * G * cin, tag=f"max{cin}")
        else:
            out = _empty((P, cin), dev)
            arg = ymax = None
            _lib.call("pcl_bn_act_f32", _P(cur), _P(in_scale), _P(in_shift), out_slope, P, cin, _P(out), st)
        ctx.x_grad_from = x_grad_from
        ctx.link = link
        ctx.cfg = (ns, slope, out_slope, training, bn, L, P, C0)
        # Everything goes through save_for_backward (never as plain ctx attributes): `out` is an OUTPUT of this node,
        # and an attribute reference to it would form a node <-> tensor cycle that only the cyclic GC frees -- with
        # ~2 GB of activations per step that is a leak of the whole working set every iteration.
        pack = _Pack()
        pack.add("x", x); pack.add_list("Ys", Ys); pack.add_list("scales", scales); pack.add_list("shifts", shifts)
        pack.add_list("means", means); pack.add_list("invstds", invstds); pack.add("out", out); pack.add("arg", arg)
        pack.add("ymax", ymax); pack.add_list("params", list(params)); pack.add("W0p", W0p)
        if rowset is not None:
            pack.add("row_meta", rowset.row_meta); pack.add("group_off", rowset.group_off)
            ctx.rowdims = (rowset.B, rowset.m, rowset.ns)
        else:
            ctx.rowdims = None
        ctx.layout = pack.layout
        ctx.save_for_backward(*pack.tensors)
        return out

    @staticmethod
    def backward(ctx, gout):
        ns, slope, out_slope, training, bn, L, P, C0 = ctx.cfg
        x_grad_from = ctx.x_grad_from
        sv = _Pack.unpack(ctx.layout, ctx.saved_tensors)
        rmeta = nrows = None
        if ctx.rowdims is not None:
            rmeta = sv["row_meta"]
            nrows = sv["group_off"][ctx.rowdims[0] * ctx.rowdims[1]:]
        params = sv["params"]
        x, Ys, scales, shifts, means, invstds = sv["x"], sv["Ys"], sv["scales"], sv["shifts"], sv["means"], sv["invstds"]
        out, arg, ymax = sv["out"], sv["arg"], sv["ymax"]
        dev = x.device
        st = _stream()
        lib = _lib.lib()
        gout = gout.contiguous()
        rows_c = ctypes.c_int(0)
        cl = Ys[-1].shape[1]
        stats = _empty((1024, 2, cl), dev, torch.float64)
        if ns:
            G = P // ns
            gz = _empty((G, cl), dev)
            _lib.call("pcl_maxgrad_prep_f32", _P(gout), _P(out), _P(ymax), out_slope, G, cl, _P(gz), _P(stats),
                      ctypes.byref(rows_c), st)
            dU, sparse = None, True
        else:
            dU = _empty((P, cl), dev)
            _lib.call("pcl_bn_act_bwd_f32", _P(gout), _P(Ys[-1]), _P(scales[-1]), _P(shifts[-1]), out_slope, P, cl, _P(dU),
                      _P(stats), ctypes.byref(rows_c), st)
            gz, sparse = None, False
        rows = rows_c.value
        grads = [None] * (6 * L)
        need_x = ctx.needs_input_grad[0]
        gx = None
        pre = None          # BatchNorm-backward constants of layer l, when the layer above already computed them
        for l in range(L - 1, -1, -1):
            W, bias, gamma, beta, _, _ = params[6 * l:6 * l + 6]
            fan_in = W.shape[1]
            if l == 0 and ctx.link is not None:
                W = W[:, :0].new_empty((W.shape[0], W.shape[0]))      # folded layer: only the shape (cout) is used below
            if l == 0 and sv["W0p"] is not None:
                W = sv["W0p"]
            cout, cin = W.shape
            if pre is not None:
                # (the layer above ran the fused kernel: its second launch already turned the sums into these constants)
                a, k1, k2, dgamma, dbeta, dbias = pre
                pre = None
                grads[6 * l + 1], grads[6 * l + 2], grads[6 * l + 3] = dbias, dgamma, dbeta
            elif bn and training:
                a, k1, k2 = _empty((3, cout), dev).unbind(0)
                dgamma, dbeta = _empty((cout,), dev), _empty((cout,), dev)
                dbias = _empty((cout,), dev) if bias is not None else None        # exactly zero under BatchNorm: cleared there
                _lib.call("pcl_bn_bwd_consts_f32", _P(stats), rows, _P(gamma), _P(means[l]), _P(invstds[l]), P, cout,
                          _P(dgamma), _P(dbeta), _P(a), _P(k1), _P(k2), _P(dbias), st)
                if syncbn.active():
                    # dgamma / dbeta above: sums over THIS rank's rows (the gradient all-reduce averages them); the constants
                    # of dy = a*du - k1 - k2*(y - mean) come from the global sums over the global row count (syncbn.py)
                    g_stats, g_rows, g_P = syncbn.reduce_rows(stats, rows, P)
                    _lib.call("pcl_bn_bwd_consts_f32", _P(g_stats), g_rows, _P(gamma), _P(means[l]), _P(invstds[l]), g_P, cout,
                              None, None, _P(a), _P(k1), _P(k2), None, st)
                grads[6 * l + 1], grads[6 * l + 2], grads[6 * l + 3] = dbias, dgamma, dbeta
            else:
                a, k1, k2 = _empty((3, cout), dev).unbind(0)
                s = stats[:rows].sum(0)
                a.copy_(scales[l]); k1.zero_(); k2.zero_()
                if bn:   # eval-mode BatchNorm: affine with constant statistics
                    grads[6 * l + 3] = s[0].float()
                    grads[6 * l + 2] = ((s[1] - means[l].double() * s[0]) * invstds[l].double()).float()
            if bias is not None and not (bn and training) and grads[6 * l + 1] is None:
                grads[6 * l + 1] = stats[:rows, 0].sum(0).float() * a             # without BatchNorm statistics: sum(du)
            if l == 0 and ctx.link is not None:
                # hand the BatchNorm-backward constants of the folded first layer to _GroupLinear.backward, which forms
                # dy = a*du - w*(k1 + k2*(y - mean)) itself; what flows back as "the gradient of x" is du (protocol
                # between the two private autograd nodes, see grouped_mlp)
                ctx.link.consts = (a, k1, k2, means[0])
                gx = dU
                break
            Xprev = Ys[l - 1] if l > 0 else x
            psc = scales[l - 1] if l > 0 else None
            psh = shifts[l - 1] if l > 0 else None
            if l > 0 and cin == fan_in and _FUSED_BWD and _lib.size_query("pcl_linear_bwd_fused_supported", cout, cin):
                # one pass forms dy once and produces BOTH the previous layer's du (+ its BatchNorm-backward sums) and dW
                nbytes = _lib.size_query("pcl_linear_bwd_fused_workspace_bytes", P, cout, cin)
                ws = _empty(((nbytes + 3) // 4,), dev)
                dW, dUp = _empty((cout, cin), dev), _empty((P, cin), dev)
                rows_n = _lib.size_query("pcl_linear_bwd_fused_stat_rows", P, cin)
                stats_n = _empty((rows_n, 2, cin), dev, torch.float64)
                per_row = 4 * (2 * cin + (cout if sparse else 2 * cout))
                _lib.call("pcl_linear_bwd_fused_rows_f32", _P(dU), _P(Ys[l]), _P(a), _P(k1), _P(k2), _P(means[l]),
                          _P(arg) if sparse else None, _P(gz) if sparse else None, ns or 1, _P(W), P, cout, cin, _P(Xprev), _P(psc),
                          _P(psh), slope, _P(dUp), _P(stats_n), _P(ws), nbytes, _P(rmeta), _P(nrows), st,
                          algo_bytes=_rows_cost(nrows, P, per_row, 8 * cin * cout),
                          algo_flops=_rows_cost(nrows, P, 4 * cin * cout, 0), tag=f"fb{cout}x{cin}")
                # second launch: partial tiles -> dW and (training BatchNorm below, statistics local to this rank) the
                # constants of layer l - 1 from the sums the fused kernel left
                if bn and training and not syncbn.active():
                    gprev, bprev = params[6 * (l - 1) + 2], params[6 * (l - 1) + 1]
                    pa, pk1, pk2, pdg, pdb = _empty((5, cin), dev).unbind(0)
                    pbias = _empty((cin,), dev) if bprev is not None else None
                    _lib.call("pcl_linear_bwd_fused_finish_f32", _P(ws), nbytes, P, cout, cin, _P(dW), _P(stats_n), _P(gprev),
                              _P(means[l - 1]), _P(invstds[l - 1]), P, _P(pdg), _P(pdb), _P(pa), _P(pk1), _P(pk2), _P(pbias), st)
                    pre = (pa, pk1, pk2, pdg, pdb, pbias)
                else:
                    _lib.call("pcl_linear_bwd_fused_finish_f32", _P(ws), nbytes, P, cout, cin, _P(dW), None, None, None, None, 0,
                              None, None, None, None, None, None, st)
                grads[6 * l] = dW
                dU, sparse, stats, rows = dUp, False, stats_n, rows_n
                continue
            nbytes = _lib.size_query("pcl_linear_bwd_dw_workspace_bytes", P, cout, cin)
            ws = _empty(((nbytes + 3) // 4,), dev)
            dW = _empty((cout, cin), dev)
            _lib.call("pcl_linear_bwd_dw_rows_f32", _P(dU), _P(Ys[l]), _P(a), _P(k1), _P(k2), _P(means[l]), _P(arg) if sparse else None,
                      _P(gz) if sparse else None, ns or 1, _P(Xprev), _P(psc), _P(psh), slope, P, cout, cin, _P(dW), _P(ws),
                      nbytes, _P(rmeta), _P(nrows), 0, st,
                      algo_bytes=_rows_cost(nrows, P, 4 * (cin + (cout if sparse else 2 * cout)), 4 * cin * cout),
                      algo_flops=_rows_cost(nrows, P, 2 * cin * cout, 0), tag=f"dw{cout}x{cin}")
            grads[6 * l] = dW if cin == fan_in else dW[:, :fan_in].contiguous()
            if l > 0 or need_x:
                dUp = _empty((P, cin), dev)
                if l > 0:
                    rows_n = _lib.size_query("pcl_mlp_stat_rows", P, cin, 1 | (2 if rmeta is not None else 0))
                    stats_n = _empty((rows_n, 2, cin), dev, torch.float64)
                else:
                    rows_n, stats_n = rows, None
                # zero-padded input rows (cin > fan_in): only the fan_in real columns are computed, the output keeps the
                # padded row stride (the pad columns are never read: the scatter takes columns [x_grad_from, fan_in))
                _lib.call("pcl_linear_bwd_dx_rows_f32", _P(dU), _P(Ys[l]), _P(a), _P(k1), _P(k2), _P(means[l]), _P(arg) if sparse else None,
                          _P(gz) if sparse else None, ns or 1, _P(W), P, cout, fan_in, _P(Xprev) if l > 0 else None, _P(psc),
                          _P(psh), slope, _P(dUp), _P(stats_n), _P(rmeta), _P(nrows), x_grad_from if l == 0 else 0,
                          cin if cin != fan_in else 0, st,
                          algo_bytes=_rows_cost(nrows, P, 4 * (cin * (2 if l > 0 else 1) + (cout if sparse else 2 * cout)), 4 * cin * cout),
                          algo_flops=_rows_cost(nrows, P, 2 * cin * cout, 0), tag=f"dx{cout}x{cin}")
                dU, sparse, stats, rows = dUp, False, stats_n, rows_n
                if l == 0:
                    gx = dUp
        return (gx, None) + tuple(grads)


def pointwise_mlp(module, x, group_max=None, rowset=None, x_grad_from=0):
    """Run ``PointwiseMLP`` ``module`` on channel-last ``x`` [..., C0] through the fused HIP path.  With a ``RowSet``
    (duplicate-compacted ball-query groups) ``x`` is the [capacity, C0] row table and the result is [B, m, CL]."""
    if not x.is_cuda:
        raise RuntimeError("fused HIP MLP needs GPU tensors (no CPU fallback)")
    if x.dtype != torch.float32:
        raise TypeError(f"expected float32, got {x.dtype}")
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    ns = 0
    if rowset is not None:
        ns = rowset.ns
    elif group_max is not None:
        assert x.shape[-2] == group_max, f"group_max={group_max} but group axis is {x.shape[-2]}"
        ns = int(group_max)
    params = []
    for i in range(module.n_layers):
        params += [module.weights[i], None if module.biases is None else module.biases[i],
                   module.gammas[i] if module.bn else None, module.betas[i] if module.bn else None,
                   getattr(module, f"running_mean_{i}") if module.bn else None,
                   getattr(module, f"running_var_{i}") if module.bn else None]
    # x_grad_from: the first input column whose gradient anybody consumes (3 for a grouped [xyz | features] tensor:
    # xyz never needs a gradient); lower columns of the returned input gradient are left unwritten.
    cfg = (ns, module.slope, module.eps, module.momentum, module.training, module.bn, module.last_act, rowset,
           int(x_grad_from), None)
    out = _FusedMLP.apply(x2, cfg, *params)
    if rowset is not None:
        return out.reshape(rowset.B, rowset.m, out.shape[-1])
    if ns:
        return out.reshape(*lead[:-1], out.shape[-1])
    return out.reshape(*lead, out.shape[-1])


class _Link:
    """Side channel between _GroupLinear and _FusedMLP (both private): forward hands over the BatchNorm partial sums of the
    folded first layer, backward hands back its BatchNorm-backward constants."""
    __slots__ = ("stats", "rows", "consts")

    def __init__(self):
        self.stats = self.rows = self.consts = None


_CONST = {}


def _const_vec(dev, n, value):
    """cached per-device constant vectors (ones / zeros) for the plain-GEMM use of the BatchNorm-backward entry points"""
    key = (dev, n, value)
    if key not in _CONST:
        _CONST[key] = torch.full((n,), float(value), dtype=torch.float32, device=dev)
    return _CONST[key]


class _GroupLinear(torch.autograd.Function):
    """Pre-BatchNorm output of the first MLP layer for the distinct rows of every ball-query group
    (csrc/compact.hip, group_linear_kernel): ``y = W0[:, :3] (xyz_nbr - centre) + W0[:, 3:] feat_nbr``.  Features of up to 4
    columns (the normals of the first level) are folded inline; wider ones go through the per-point product
    ``Uf = feat W0[:, 3:]^T`` -- one GEMM over the N points of a cloud instead of its m*ns grouped rows -- on the library's
    own GEMM kernels, forward and backward."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, feature, W0, idx, cnt, group_off, link, use_xyz):
        B, m, ns = idx.shape
        N = xyz.shape[1]
        C1 = W0.shape[0]
        off = 3 if use_xyz else 0
        C = 0 if feature is None else feature.shape[-1]
        dev = idx.device
        cap = B * m * ns
        st = _stream()
        inline = 0 < C <= 4 and not feature.requires_grad
        W0 = W0.contiguous()
        Wx = W0[:, :3] if use_xyz else None                    # views: the kernels take the stored weight with its row stride
        Wf = (W0[:, off:] if inline else W0[:, off:].contiguous()) if C else None       # the point GEMM wants it dense
        feat2 = feature.reshape(B * N, C).contiguous() if C else None
        Uf = None
        if C and not inline:
            Uf = _empty((B * N, C1), dev)
            rows_u = _lib.size_query("pcl_mlp_stat_rows", B * N, C1, 0)
            _lib.call("pcl_linear_fwd_rows_f32", _P(feat2), _P(Wf), None, None, None, 0.0, B * N, C, C1, _P(Uf),
                      _P(_empty((rows_u, 2, C1), dev, torch.float64)), None, None, st, tag=f"pt{C}x{C1}")
        Y = _empty((cap, C1), dev)
        row_meta = _empty((cap, 2), dev, torch.int32)
        row_src = _empty((cap,), dev, torch.int32)
        rows = _lib.size_query("pcl_group_linear_stat_rows", B, m)
        stats = _empty((rows, 2, C1), dev, torch.float64)
        row_loc = _empty((cap, 4), dev)
        row_feat = _empty((cap, 4), dev) if inline else None
        _lib.call("pcl_group_linear_f32", _P(xyz), _P(new_xyz), _P(Uf), _P(Wx), _P(feat2) if inline else None,
                  _P(Wf) if inline else None, C if inline else 0, W0.shape[1], _P(idx), _P(cnt), _P(group_off), B, N, m, ns, C1, _P(Y),
                  _P(row_meta), _P(row_src), _P(row_loc), _P(row_feat), _P(stats), st,
                  # per distinct row: Y written, Uf row gathered (L2), 16-byte records; idx read once
                  algo_bytes=_rows_cost(group_off[B * m:], cap, 4 * C1 * (2 if Uf is not None else 1) + 36, 4 * cap), tag=f"glin{C1}")
        link.stats, link.rows = stats, rows
        ctx.link = link
        ctx.dims = (B, N, m, ns, C1, C, off, inline, rows)
        ctx.mark_non_differentiable(row_meta, row_src)
        ctx.set_materialize_grads(False)          # no zero tensors for the (integer) metadata outputs
        ctx.save_for_backward(group_off, Y, row_src, row_loc, row_feat, feat2, None if inline else Wf)
        return Y, row_meta, row_src

    @staticmethod
    def backward(ctx, du, *_):
        B, N, m, ns, C1, C, off, inline, rows = ctx.dims
        group_off, Y, row_src, row_loc, row_feat, feat2, Wf = ctx.saved_tensors
        a, k1, k2, mu = ctx.link.consts
        ctx.link.consts = None
        dev = Y.device
        st = _stream()
        lib = _lib.lib()
        if du is None:
            return (None,) * 9
        need_w = ctx.needs_input_grad[3]
        wide = C > 0 and not inline
        fan_in = off + C
        dUf = _empty((B * N, C1), dev) if wide else None
        dWxp = _empty((rows, C1, 3), dev) if (off and need_w) else None
        dWfp = _empty((rows, C1, C), dev) if (inline and need_w) else None
        dW0 = _empty((C1, fan_in), dev) if need_w else None
        if dUf is not None or dWxp is not None or dWfp is not None:
            fin = dW0 is not None and (dWxp is not None or dWfp is not None)       # partial sums -> dW0 columns, in the same call
            _lib.call("pcl_group_linear_bwd_f32", _P(row_loc), _P(row_feat), C if inline else 0, _P(du.contiguous()), _P(Y),
                      _P(a), _P(k1), _P(k2), _P(mu), _P(row_src), _P(group_off[B * m:]), B, N, C1, _P(dUf), _P(dWxp), _P(dWfp),
                      _P(dW0) if fin else None, fan_in, off, st,
                      algo_bytes=_rows_cost(group_off[B * m:], B * m * ns, 4 * C1 * (3 if wide else 2) + 36, 0), tag=f"glinbwd{C1}")
        dfeat = None
        if wide:
            # plain GEMMs through the BatchNorm-backward entry points with a = 1, k1 = k2 = 0 (dy == dUf)
            one, zero = _const_vec(dev, C1, 1.0), _const_vec(dev, C1, 0.0)
            P = B * N
            if need_w:
                nbytes = _lib.size_query("pcl_linear_bwd_dw_workspace_bytes", P, C1, C)
                ws = _empty(((nbytes + 3) // 4,), dev)
                _lib.call("pcl_linear_bwd_dw_rows_f32", _P(dUf), _P(dUf), _P(one), _P(zero), _P(zero), _P(zero), None, None, 1,
                          _P(feat2), None, None, 0.0, P, C1, C, _P(dW0[:, off:]), _P(ws), nbytes, None, None, fan_in, st,
                          tag=f"ptdw{C1}x{C}")
            if ctx.needs_input_grad[2]:
                dfeat = _empty((P, C), dev)
                _lib.call("pcl_linear_bwd_dx_rows_f32", _P(dUf), _P(dUf), _P(one), _P(zero), _P(zero), _P(zero), None, None, 1,
                          _P(Wf), P, C1, C, None, None, None, 0.0, _P(dfeat), None, None, None, 0, 0, st, tag=f"ptdx{C1}x{C}")
                dfeat = dfeat.view(B, N, C)
        return None, None, dfeat, dW0, None, None, None, None, None


def can_fold_first_layer(module, use_xyz, feature):
    """The folded path needs >= 2 layers (the first layer's gradient arrives dense from the second), no conv bias and at
    most 256 first-layer outputs."""
    return (module.n_layers >= 2 and module.biases is None and module.spec[1] <= 256 and (use_xyz or feature is not None))


def grouped_mlp(module, xyz, new_xyz, feature, idx, cnt, group_off, use_xyz):
    """Ball-query grouping + ``PointwiseMLP`` + max over each group, with the first conv folded into the grouping:
    ``W0 [xyz_nbr - centre | feat_nbr] = W0[:, :3] (xyz_nbr - centre) + (feat W0[:, 3:]^T)[nbr]`` -- the feature product
    is one GEMM over the N points of a cloud (plain library GEMM) instead of over its m*ns grouped rows.  Returns
    [B, m, C_last]."""
    from .ops import RowSet
    if not xyz.is_cuda:
        raise RuntimeError("fused HIP MLP needs GPU tensors (no CPU fallback)")
    B, m, ns = idx.shape
    link = _Link()
    Y0, row_meta, row_src = _GroupLinear.apply(xyz.contiguous(), new_xyz.contiguous(), feature, module.weights[0], idx, cnt,
                                               group_off, link, bool(use_xyz))
    rowset = RowSet(B, m, ns, row_meta, row_src, group_off)
    params = []
    for i in range(module.n_layers):
        params += [module.weights[i] if i > 0 else None, None,
                   module.gammas[i] if module.bn else None, module.betas[i] if module.bn else None,
                   getattr(module, f"running_mean_{i}") if module.bn else None,
                   getattr(module, f"running_var_{i}") if module.bn else None]
    params[0] = module.weights[0].detach()          # shape carrier only: the folded layer's weight gets its gradient outside
    cfg = (ns, module.slope, module.eps, module.momentum, module.training, module.bn, module.last_act, rowset, 0, link)
    out = _FusedMLP.apply(Y0, cfg, *params)
    return out.reshape(B, m, out.shape[-1])
