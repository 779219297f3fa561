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


_UNIT_CONSTS = {}


def _unit_consts(dev, c):
    """(ones[c], zeros[c]) on ``dev``, cached: the identity scale / zero shift of a layer without BatchNorm (never written to)."""
    key = (dev.type, dev.index, c)
    v = _UNIT_CONSTS.get(key)
    if v is None:
        v = _UNIT_CONSTS[key] = (torch.ones(c, device=dev), torch.zeros(c, device=dev))
    return v


class _FusedMLP(torch.autograd.Function):
    """inputs: x [P,C0]; cfg = (group_ns|0, slope, eps, momentum, training, bn, last_act); then per layer
    (W, bias|None, gamma|None, beta|None, running_mean|None, running_var|None)."""

    @staticmethod
    def forward(ctx, x, cfg, *params):
        ns, slope, eps, momentum, training, bn, last_act, rowset, x_grad_from, link = cfg[:10]
        flush_k = cfg[10] if len(cfg) > 10 else 0          # forward GEMMs on plain rows: fp64-flushed accumulation (csrc/frag.hip)
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
            fused_max = (l == L - 1) and ns in (32, 64) and rowset is None and not flush_k
            folded = l == 0 and link is not None
            flushed = bool(flush_k) and rowset is None and not folded
            if folded:
                # the first layer was folded into the grouping (_GroupLinear): x IS its pre-BatchNorm output
                assert cout == cin and L >= 2, "folded first layer: x must be the layer's own output"
                Y, stats, rows = x, link.stats, link.rows
            else:
                Y = _empty((P, cout), dev)
                rows = _lib.size_query("pcl_frag_stat_rows", P) if flushed else _lib.size_query("pcl_mlp_stat_rows", P, cout, 2 if rowset is not None else 0)
                stats = _empty((rows, 2, cout), dev, torch.float64)
            if folded:
                pass                       # nothing to launch: the grouping kernel already produced Y and its sums
            elif flushed:
                _lib.call("pcl_frag_linear_fwd_f32", _P(cur), cin, _P(W), cin, _P(bias), _P(in_scale), _P(in_shift), slope, P, cin, cout, _P(Y), cout,
                          None, _P(stats), flush_k, st, algo_bytes=4 * P * (cin + cout) + 4 * cin * cout, algo_flops=2 * P * cin * cout, tag=f"fwd{cin}x{cout}")
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
                scale, shift = _unit_consts(dev, cout)      # cached constants (read-only everywhere): no fill launches per call
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
            fewrow = False
            if pre is not None:
                # (the layer above ran the fused kernel: its second launch already turned the sums into these constants)
                a, k1, k2, dgamma, dbeta, dbias = pre
                pre = None
                grads[6 * l + 1], grads[6 * l + 2], grads[6 * l + 3] = dbias, dgamma, dbeta
            elif bn and training:
                a, k1, k2 = _empty((3, cout), dev).unbind(0)
                dgamma, dbeta = _empty((cout,), dev), _empty((cout,), dev)
                dbias = _empty((cout,), dev) if bias is not None else None        # exactly zero under BatchNorm: cleared there
                # few-row layers (csrc/stack.hip: fewrow_layer -- the same question, the same kernels): constants AND dy in one launch
                fewrow = (rmeta is None and ctx.link is None and cin == fan_in and not syncbn.active()
                          and bool(_lib.size_query("pcl_mlp_fewrow_layer", P, cout, cin, int(l == 0))))
                if fewrow:
                    dy = _empty((P, cout), dev)
                    _lib.call("pcl_bn_bwd_dy_f32", _P(stats), rows, _P(gamma), _P(means[l]), _P(invstds[l]), P, cout, _P(dgamma), _P(dbeta), _P(a),
                              _P(k1), _P(k2), _P(dbias), None if sparse else _P(dU), _P(Ys[l]), _P(arg) if sparse else None,
                              _P(gz) if sparse else None, ns or 1, P, _P(dy), st)
                else:
                    _lib.call("pcl_bn_bwd_consts_f32", _P(stats), rows, _P(gamma), _P(means[l]), _P(invstds[l]), P, cout,
                              _P(dgamma), _P(dbeta), _P(a), _P(k1), _P(k2), _P(dbias), st)
                if syncbn.active():
                    # dgamma / dbeta above: sums over THIS rank's rows (the gradient all-reduce averages them); the constants
                    # of dy = a*du - k1 - k2*(y - mean) come from the global sums over the global row count (syncbn.py)
                    g_stats, g_rows, g_P = syncbn.reduce_rows(stats, rows, P)
                    _lib.call("pcl_bn_bwd_consts_f32", _P(g_stats), g_rows, _P(gamma), _P(means[l]), _P(invstds[l]), g_P, cout,
                              None, None, _P(a), _P(k1), _P(k2), None, st)
                grads[6 * l + 1], grads[6 * l + 2], grads[6 * l + 3] = dbias, dgamma, dbeta
            elif bn:     # eval-mode BatchNorm: affine with constant statistics
                a, k1, k2 = _empty((3, cout), dev).unbind(0)
                s = stats[:rows].sum(0)
                a.copy_(scales[l]); k1.zero_(); k2.zero_()
                grads[6 * l + 3] = s[0].float()
                grads[6 * l + 2] = ((s[1] - means[l].double() * s[0]) * invstds[l].double()).float()
            else:        # no BatchNorm: dy = du (a = 1, k1 = k2 = 0: the cached constants); the bias gradient is sum(du) = what
                         # pcl_bn_bwd_consts_f32 returns as dbeta -- one launch instead of six torch kernels per layer
                a, k1 = _unit_consts(dev, cout)
                k2 = k1
                if bias is not None:
                    dbias = _empty((cout,), dev)
                    scratch = _empty((3, cout), dev)
                    _lib.call("pcl_bn_bwd_consts_f32", _P(stats), rows, None, _P(k1), _P(a), P, cout, None, _P(dbias),
                              _P(scratch[0]), _P(scratch[1]), _P(scratch[2]), None, st)
                    grads[6 * l + 1] = dbias
            if bias is not None and not (bn and training) and grads[6 * l + 1] is None:
                grads[6 * l + 1] = stats[:rows, 0].sum(0).float() * a             # eval-mode BatchNorm: sum(du) * scale
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
            if fewrow:
                nbytes = _lib.size_query("pcl_linear_bwd_dw_plain_workspace_bytes", P, cout, cin)
                ws = _empty(((nbytes + 3) // 4,), dev)
                dW = _empty((cout, cin), dev)
                _lib.call("pcl_linear_bwd_dw_plain_f32", _P(dy), _P(Xprev), _P(psc), _P(psh), slope, P, cout, cin, _P(dW), _P(ws), nbytes, 0, st,
                          algo_bytes=4 * P * (cin + cout) + 4 * cin * cout, algo_flops=2 * P * cin * cout, tag=f"dw{cout}x{cin}")
                grads[6 * l] = dW
                if l > 0 or need_x:
                    dUp = _empty((P, cin), dev)
                    rows_n = _lib.size_query("pcl_frag_stat_rows", P)
                    stats_n = _empty((rows_n, 2, cin), dev, torch.float64) if l > 0 else None
                    _lib.call("pcl_frag_linear_bwd_dx_f32", _P(dy), _P(W), cin, P, cout, cin, _P(Xprev) if l > 0 else None, cin,
                              _P(psc) if l > 0 else None, _P(psh) if l > 0 else None, slope, _P(dUp), cin, _P(stats_n),
                              x_grad_from if l == 0 else 0, st, algo_bytes=4 * P * (cin * (2 if l > 0 else 1) + cout) + 4 * cin * cout,
                              algo_flops=2 * P * cin * cout, tag=f"dx{cout}x{cin}")
                    if l > 0:
                        dU, sparse, stats, rows = dUp, False, stats_n, rows_n
                    else:
                        gx = dUp
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


def _profiling_pad(module, P, params, folded_first=False):
    """The per-stack entry points run a 96-wide hidden layer zero-padded to 128 channels (csrc/stack.hip: padded_width -- the
    resident-weight forward and the fused backward exist for 64 / 128 / 256).  The per-kernel path is what the profiling passes time
    (``per_kernel_path``), so it has to launch the same kernels: the same padding, built here with torch ops (pad: zero weights,
    gamma = beta = 0 -> the extra channels are exactly 0 everywhere; autograd slices the gradients back).  Returns the padded
    parameter list and a closure that copies the 96 real running statistics back after the forward."""
    L = module.n_layers
    spec = module.spec
    widths = [128 if (1 <= l < L and spec[l] == 96 and P >= 32768) else spec[l] for l in range(L + 1)]
    if widths == list(spec[:L + 1]) or not (module.bn and module.training) or (folded_first and widths[1] != spec[1]):
        return params, None
    F = torch.nn.functional
    out, fix = list(params), []
    for l in range(L):
        W, bias, gamma, beta, rm, rv = params[6 * l:6 * l + 6]
        co, ci = widths[l + 1] - spec[l + 1], widths[l] - spec[l]
        if W is not None and (co or ci):
            out[6 * l] = F.pad(W, (0, ci, 0, co))
        if co:
            out[6 * l + 1] = None if bias is None else F.pad(bias, (0, co))
            out[6 * l + 2], out[6 * l + 3] = F.pad(gamma, (0, co)), F.pad(beta, (0, co))
            rmp, rvp = F.pad(rm, (0, co)), F.pad(rv, (0, co), value=1.0)
            out[6 * l + 4], out[6 * l + 5] = rmp, rvp
            fix.append((rm, rmp, rv, rvp, spec[l + 1]))
    def finish():
        with torch.no_grad():
            for rm, rmp, rv, rvp, c in fix:
                rm.copy_(rmp[:c]); rv.copy_(rvp[:c])
    return out, finish


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
    if rowset is None and _stack_eligible(module) and module.weights[0].shape[1] == x2.shape[1]:
        out = stack_plain(module, x2.contiguous(), ns, int(x_grad_from))
        return out.reshape(*lead[:-1], out.shape[-1]) if ns else out.reshape(*lead, out.shape[-1])
    params = []
    for i in range(module.n_layers):
        params += [module.weights[i], None if module.biases is None else module.biases[i],
                   module.gammas[i] if module.bn else None, module.betas[i] if module.bn else None,
                   getattr(module, f"running_mean_{i}") if module.bn else None,
                   getattr(module, f"running_var_{i}") if module.bn else None]
    # x_grad_from: the first input column whose gradient anybody consumes (3 for a grouped [xyz | features] tensor:
    # xyz never needs a gradient); lower columns of the returned input gradient are left unwritten.
    cfg = (ns, module.slope, module.eps, module.momentum, module.training, module.bn, module.last_act, rowset,
           int(x_grad_from), None, int(getattr(module, "flush_k", 0)) if rowset is None else 0)
    params, finish = _profiling_pad(module, x2.shape[0], params) if rowset is None else (params, None)
    out = _FusedMLP.apply(x2, cfg, *params)
    if finish is not None:
        finish()
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
        # wide features: the backward's scatter runs as a gather over every source point's rows (what the per-stack entry point does)
        in_off = in_rows = None
        if (Uf is not None and _lib.size_query("pcl_group_linear_bwd_gather_supported", C1)
                and _lib.size_query("pcl_group_rows_transpose_supported", N, m, ns)):
            in_off, in_rows = _empty((B * N + 1,), dev, torch.int32), _empty((cap,), dev, torch.int32)
            _lib.call("pcl_group_rows_transpose_i32", _P(row_src), _P(group_off), B, N, m, ns, _P(in_off), _P(in_rows), st)
        ctx.link = link
        ctx.dims = (B, N, m, ns, C1, C, off, inline, rows)
        ctx.mark_non_differentiable(row_meta, row_src)
        ctx.set_materialize_grads(False)          # no zero tensors for the (integer) metadata outputs
        ctx.save_for_backward(group_off, Y, row_src, row_loc, row_feat, feat2, None if inline else Wf, in_off, in_rows)
        return Y, row_meta, row_src

    @staticmethod
    def backward(ctx, du, *_):
        B, N, m, ns, C1, C, off, inline, rows = ctx.dims
        group_off, Y, row_src, row_loc, row_feat, feat2, Wf, in_off, in_rows = ctx.saved_tensors
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
        if dUf is not None and in_off is not None:
            _lib.call("pcl_group_linear_bwd_gather_f32", _P(row_loc), _P(du.contiguous()), _P(Y), _P(a), _P(k1), _P(k2), _P(mu), _P(in_off), _P(in_rows),
                      B, N, C1, _P(dUf), _P(dWxp), _P(dW0) if (dW0 is not None and dWxp is not None) else None, fan_in, st,
                      algo_bytes=_rows_cost(group_off[B * m:], B * m * ns, 4 * C1 * 2 + 36, 4 * B * N * C1), tag=f"glinbwd{C1}")
        elif dUf is not None or dWxp is not None or dWfp is not None:
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
    if _stack_eligible(module):
        return stack_grouped(module, xyz.contiguous(), new_xyz.contiguous(), feature, idx, cnt, group_off, use_xyz)
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
    params, finish = _profiling_pad(module, B * m * ns, params, folded_first=True)
    out = _FusedMLP.apply(Y0, cfg, *params)
    if finish is not None:
        finish()
    return out.reshape(B, m, out.shape[-1])


# ------------------------------------------------------------------------------------------------------------------------
# Per-stack entry points (csrc/stack.hip): ONE C-ABI call runs a whole stack forward or backward -- the unit the reference
# runs per module ``execute`` (networks/cls/pointnet2.py:33-62).  Same kernels, same order, same results as the per-kernel
# path above (tests/test_mlp_hip.py::test_stack_entry_points_*); what changes is the host side: one descriptor, one
# persistent + one transient buffer, one autograd node.  Eligible: training-mode BatchNorm on every layer with
# process-local statistics; anything else (evaluation mode, SyncBN, bn=False, zero-padded input rows) keeps the path above.
USE_STACK = os.environ.get("PCL_STACK", "1") != "0"
_MAXL = 8


class _CLayer(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("W", "bias", "gamma", "beta", "running_mean", "running_var", "dW", "dbias",
                                               "dgamma", "dbeta")]


_GOUT_IN_PLACE = __import__("os").environ.get("PCL_GOUT_IN_PLACE", "1") != "0"      # lab switch (A/B on one box)
COUNTERS = {"strided_gout": 0}        # calls of the stack backward that read their gout in place from a wider gradient


class _CStack(ctypes.Structure):
    _fields_ = ([("struct_bytes", ctypes.c_int32), ("n_layers", ctypes.c_int32), ("c", ctypes.c_int32 * (_MAXL + 1))]
                + [(n, ctypes.c_int32) for n in ("P", "pool", "grouped", "x_grad_from", "need_dx", "B", "N", "m", "Cf", "use_xyz")]
                + [(n, ctypes.c_float) for n in ("slope", "out_slope", "eps", "momentum")]
                + [(n, ctypes.c_void_p) for n in ("x", "xyz", "new_xyz", "feature", "Wf_dense", "idx", "cnt", "group_off")]
                + [("layer", _CLayer * _MAXL), ("out", ctypes.c_void_p), ("save", ctypes.c_void_p), ("save_bytes", ctypes.c_size_t),
                   ("tmp", ctypes.c_void_p), ("tmp_bytes", ctypes.c_size_t), ("gout", ctypes.c_void_p), ("dx", ctypes.c_void_p),
                   ("stream", ctypes.c_void_p), ("defer_act", ctypes.c_int32), ("ext_stat_rows", ctypes.c_int32),
                   ("ext_stats", ctypes.c_void_p), ("flush_k", ctypes.c_int32), ("gout_ld", ctypes.c_int32)])


class _StackPlan:
    """Everything about one (module, input shape) that does not change from step to step: the filled-in descriptor (only
    pointers are patched per call), buffer sizes, the layout of the flat parameter-gradient buffer."""
    __slots__ = ("desc", "ref", "save_bytes", "fwd_tmp", "bwd_tmp", "G", "cl", "L", "has_bias", "gsizes", "gshapes", "gtotal",
                 "spec", "grouped", "wide", "off", "P", "c0", "defer", "last_off", "out_slope")


_PLANS = {}


def _stack_plan(module, P, c0, pool, grouped, geom, need_dx, x_grad_from, defer=False):
    flush_k = int(getattr(module, "flush_k", 0))
    key = (id(module), P, c0, pool, grouped, geom, need_dx, x_grad_from, module.slope, module.last_act, defer,
           module.eps, module.momentum, module.biases is not None, flush_k)      # (everything the filled-in descriptor depends on)
    plan = _PLANS.get(key)
    if plan is not None and plan.spec is module.spec:
        return plan
    L = module.n_layers
    d = _CStack()
    d.struct_bytes = ctypes.sizeof(_CStack)
    d.n_layers = L
    spec = module.spec
    d.c[0] = c0
    for l in range(L):
        d.c[l + 1] = spec[l + 1]
    d.P, d.pool, d.grouped, d.x_grad_from, d.need_dx = P, pool, int(grouped), int(x_grad_from), int(need_dx)
    d.defer_act = int(defer)
    d.flush_k = flush_k
    if grouped:
        d.B, d.N, d.m, d.Cf, d.use_xyz = geom
    d.slope, d.out_slope, d.eps, d.momentum = module.slope, (module.slope if module.last_act else 1.0), module.eps, module.momentum
    # pointers that validate() wants non-null for the size query; patched for real on every call
    d.x = 1
    d.xyz = d.new_xyz = d.idx = d.cnt = d.group_off = 1
    d.feature = d.Wf_dense = 1
    for l in range(L):
        d.layer[l].W = d.layer[l].gamma = d.layer[l].beta = 1
    sv, ft, bt = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    _lib.check(_lib.lib().pcl_mlp_stack_sizes(ctypes.byref(d), ctypes.byref(sv), ctypes.byref(ft), ctypes.byref(bt)), "pcl_mlp_stack_sizes")
    plan = _StackPlan()
    plan.desc, plan.ref = d, ctypes.byref(d)
    plan.save_bytes, plan.fwd_tmp, plan.bwd_tmp = sv.value, ft.value, bt.value
    plan.defer, plan.last_off, plan.out_slope = bool(defer), None, (module.slope if module.last_act else 1.0)
    if defer:
        yo, so, ho = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
        _lib.check(_lib.lib().pcl_mlp_stack_last(ctypes.byref(d), ctypes.byref(yo), ctypes.byref(so), ctypes.byref(ho)), "pcl_mlp_stack_last")
        plan.last_off = (yo.value, so.value, ho.value)
    plan.L, plan.cl, plan.spec, plan.grouped, plan.P, plan.c0 = L, spec[L], module.spec, grouped, P, c0
    plan.G = (geom[0] * geom[2]) if grouped else (P // pool if pool else 0)
    plan.has_bias = module.biases is not None
    plan.off = 3 if (grouped and geom[4]) else 0
    plan.wide = bool(grouped and geom[3] > 0 and (geom[3] > 4 or need_dx))
    # flat parameter-gradient buffer: per layer dW [cout, cin], dgamma, dbeta (, dbias)
    sizes, shapes = [], []
    for l in range(L):
        cin, cout = (c0 if l == 0 else spec[l]), spec[l + 1]
        sizes += [cout * cin, cout, cout] + ([cout] if plan.has_bias else [])
        shapes.append((cout, cin))
    plan.gsizes, plan.gshapes, plan.gtotal = sizes, shapes, sum(sizes)
    _PLANS[key] = plan
    return plan


class per_kernel_path:
    """``with per_kernel_path(): ...`` -- run the stacks and the FC head through the per-kernel entry points (one C-ABI call per
    kernel) instead of the per-stack ones: the same kernels in the same order, bit-identical results
    (tests/test_mlp_hip.py) -- except for the narrow stacks (PointConv's WeightNet / DensityNet, csrc/narrow.hip), which exist
    behind the per-stack entry point only and run on the GEMM kernels here (equal to rounding,
    ``test_narrow_stack_equals_the_gemm_path_at_size``).  For the profiling passes that bracket individual C-ABI calls with events
    (``_lib.KernelTimer``: bench.py's choice of the dominant kernel, tools/bench_models.py, ``--profile-all``)."""

    def __enter__(self):
        global USE_STACK
        from . import head
        self.prev = (USE_STACK, head.USE_STACK)
        USE_STACK = head.USE_STACK = False

    def __exit__(self, *a):
        global USE_STACK
        from . import head
        USE_STACK, head.USE_STACK = self.prev


def _stack_eligible(module):
    return (USE_STACK and module.bn and module.training and module.n_layers <= _MAXL and not syncbn.active())


class _StackFn(torch.autograd.Function):
    """inputs: t_in = x [P, c0] (plain stack) or the point features [B, N, Cf] | None (grouped stack); ``aux`` = (plan, index
    tensors of a grouped stack | None); then per layer W, gamma, beta (, bias); running statistics travel in ``aux``."""

    @staticmethod
    def forward(ctx, t_in, aux, *params):
        plan, geo, running = aux[0], aux[1], aux[2]
        # a private copy of the plan's descriptor per call (~1 KB): forward and autograd's backward thread, or two host threads,
        # may be inside the same call site at once
        d = type(plan.desc).from_buffer_copy(plan.desc)
        L = plan.L
        dev = params[0].device
        npl = 4 if plan.has_bias else 3
        st = _stream()
        save = torch.empty((plan.save_bytes,), dtype=torch.uint8, device=dev)
        tmp = torch.empty((plan.fwd_tmp,), dtype=torch.uint8, device=dev)
        if plan.defer:
            # the consumer applies the last BatchNorm + activation itself (PointConv's contraction): what this node returns is
            # the last PRE-BatchNorm output, a view into `save`; scale / shift travel through the link
            yo, so, ho = plan.last_off
            out = save[yo:yo + 4 * plan.P * plan.cl].view(torch.float32).view(plan.P, plan.cl)
            link = aux[3]
            link.scale = save[so:so + 4 * plan.cl].view(torch.float32)
            link.shift = save[ho:ho + 4 * plan.cl].view(torch.float32)
            link.slope = plan.out_slope
        else:
            out = torch.empty((plan.G if plan.G else plan.P, plan.cl), dtype=torch.float32, device=dev)
        Wf = None
        if plan.grouped:
            xyz, new_xyz, idx, cnt, group_off = geo
            d.xyz, d.new_xyz, d.idx, d.cnt, d.group_off = xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(), cnt.data_ptr(), group_off.data_ptr()
            d.feature = None if t_in is None else t_in.data_ptr()
            if plan.wide:
                Wf = torch.empty((params[0].shape[0], params[0].shape[1] - plan.off), dtype=torch.float32, device=dev)   # filled by the call (dense feature columns of W0)
            d.Wf_dense = None if Wf is None else Wf.data_ptr()
            d.x = None
        else:
            d.x = t_in.data_ptr()
        for l in range(L):
            ly, q = d.layer[l], params[npl * l:npl * l + npl]
            ly.W, ly.gamma, ly.beta = q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr()
            ly.bias = q[3].data_ptr() if plan.has_bias else None
            ly.running_mean, ly.running_var = running[2 * l].data_ptr(), running[2 * l + 1].data_ptr()
        d.out = None if plan.defer else out.data_ptr()
        d.save, d.save_bytes, d.tmp, d.tmp_bytes, d.stream = save.data_ptr(), plan.save_bytes, tmp.data_ptr(), plan.fwd_tmp, st
        _lib.call("pcl_mlp_stack_fwd_f32", ctypes.byref(d), tag="stack_fwd")
        ctx.plan = plan
        ctx.link = aux[3] if plan.defer else None
        ctx.save_for_backward(t_in, out, save, Wf, *(geo if plan.grouped else ()), *params)
        return out

    @staticmethod
    def backward(ctx, gout):
        plan = ctx.plan
        # a private copy of the plan's descriptor per call (~1 KB): forward and autograd's backward thread, or two host threads,
        # may be inside the same call site at once
        d = type(plan.desc).from_buffer_copy(plan.desc)
        L = plan.L
        sv = ctx.saved_tensors
        t_in, out, save, Wf = sv[0], sv[1], sv[2], sv[3]
        n0 = 4
        if plan.grouped:
            xyz, new_xyz, idx, cnt, group_off = sv[4:9]
            n0 = 9
        params = sv[n0:]
        dev = out.device
        npl = 4 if plan.has_bias else 3
        # a pooled stack whose output went into a concatenation (multi-scale grouping) gets a column slice of the wide gradient:
        # the max-gradient kernel reads it in place (row stride gout_ld), no copy per scale
        d.gout_ld = 0
        if (_GOUT_IN_PLACE and plan.G and not plan.defer and gout.dim() == 2 and gout.stride(1) == 1 and gout.stride(0) > gout.shape[1]
                and gout.stride(0) < 2 ** 31 and gout.dtype == torch.float32):
            d.gout_ld = gout.stride(0)
            COUNTERS["strided_gout"] += 1
        else:
            gout = gout.contiguous()
        tmp = torch.empty((plan.bwd_tmp,), dtype=torch.uint8, device=dev)
        flat = torch.empty((plan.gtotal,), dtype=torch.float32, device=dev)
        pieces = flat.split_with_sizes(plan.gsizes)
        dx = None
        need_dx = ctx.needs_input_grad[0]
        if need_dx:
            dx = torch.empty_like(t_in)
        if plan.grouped:
            d.xyz, d.new_xyz, d.idx, d.cnt, d.group_off = xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(), cnt.data_ptr(), group_off.data_ptr()
            d.feature = None if t_in is None else t_in.data_ptr()
            d.Wf_dense = None if Wf is None else Wf.data_ptr()
            d.x = None
        else:
            d.x = t_in.data_ptr()
        base, o = flat.data_ptr(), 0
        grads = []
        for l in range(L):
            ly, q = d.layer[l], params[npl * l:npl * l + npl]
            ly.W, ly.gamma, ly.beta = q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr()
            ly.bias = q[3].data_ptr() if plan.has_bias else None
            cout, cin = plan.gshapes[l]
            k = npl * l
            ly.dW = base + 4 * o; o += cout * cin
            ly.dgamma = base + 4 * o; o += cout
            ly.dbeta = base + 4 * o; o += cout
            grads += [pieces[k].view(cout, cin), pieces[k + 1], pieces[k + 2]]
            if plan.has_bias:
                ly.dbias = base + 4 * o; o += cout
                grads.append(pieces[k + 3])
            else:
                ly.dbias = None
        d.out = None if plan.defer else out.data_ptr()
        d.save, d.save_bytes, d.tmp, d.tmp_bytes = save.data_ptr(), plan.save_bytes, tmp.data_ptr(), plan.bwd_tmp
        d.gout, d.dx, d.stream = gout.data_ptr(), (None if dx is None else dx.data_ptr()), _stream()
        if plan.defer:
            # gout IS du (the consumer masked it with the activation and left the BatchNorm-backward sums in the link)
            link = ctx.link
            d.ext_stats, d.ext_stat_rows = link.stats.data_ptr(), link.rows
            link.stats = None
        _lib.call("pcl_mlp_stack_bwd_f32", ctypes.byref(d), tag="stack_bwd")
        return (dx, None) + tuple(grads)


def _stack_params(module):
    cached = module.__dict__.get("_stack_params_cache")
    if (cached is not None and cached[0][0] is module.weights[0] and cached[1][0] is module._buffers.get("running_mean_0")
            and cached[0][-1] is (module.biases[-1] if module.biases is not None else module.betas[-1]) and cached[1][-1] is module._buffers.get(f"running_var_{module.n_layers - 1}")):
        return cached
    ps, running = [], []
    for i in range(module.n_layers):
        ps += [module.weights[i], module.gammas[i], module.betas[i]] + ([module.biases[i]] if module.biases is not None else [])
        running += [getattr(module, f"running_mean_{i}"), getattr(module, f"running_var_{i}")]
    module.__dict__["_stack_params_cache"] = (ps, running)      # (.to() / load_state_dict keep the Parameter objects; the identity check above catches replaced ones)
    return ps, running


def stack_plain(module, x2, ns, x_grad_from):
    """plain stack on rows x2 [P, C0] (no RowSet); returns [P, CL] or [P/ns, CL]"""
    P, c0 = x2.shape
    need_dx = x2.requires_grad
    plan = _stack_plan(module, P, c0, ns, False, None, need_dx, x_grad_from if need_dx else 0)
    ps, running = _stack_params(module)
    return _StackFn.apply(x2, (plan, None, running), *ps)


class DeferLink:
    """Side channel between a deferring stack (_StackFn, plan.defer) and the consumer that applies its last BatchNorm + activation
    (pointconv_utils._PointConvContractBN): forward hands over scale / shift / slope, backward the (sum du, sum du*y) rows."""
    __slots__ = ("scale", "shift", "slope", "stats", "rows")

    def __init__(self):
        self.scale = self.shift = self.stats = None
        self.slope, self.rows = 0.0, 0


def stack_plain_deferred(module, x2):
    """Plain stack on rows x2 [P, C0] WITHOUT its last BatchNorm + activation: returns (Y_last [P, CL] pre-BatchNorm, link) for a
    consumer that folds them into its own operand load; None when the stack path does not apply."""
    if not (_stack_eligible(module) and module.weights[0].shape[1] == x2.shape[1]):
        return None
    P, c0 = x2.shape
    need_dx = x2.requires_grad
    plan = _stack_plan(module, P, c0, 0, False, None, need_dx, 0, defer=True)
    ps, running = _stack_params(module)
    link = DeferLink()
    return _StackFn.apply(x2, (plan, None, running, link), *ps), link


_FULL = {}


def stack_grouped_deferred(module, xyz, new_xyz, feature, idx, use_xyz=True):
    """Grouping by FULL neighbour lists (k-NN groups: every slot a distinct point; PointConv's sample_and_group,
    misc/pointconv_utils.py:133-170) + ``module`` WITHOUT its last BatchNorm + activation and without a max: the first conv is
    folded into the grouping (pcl_group_linear_f32: the [B,S,ns,3+D] grouped tensor never exists, its K = 3+D first GEMM
    runs over the N points instead of the S*ns rows), the consumer applies the last BatchNorm + activation while loading.
    Returns (Y [B*S*ns, CL], link) or None when this path does not apply."""
    if feature is None or feature.shape[-1] <= 4 or not use_xyz:
        return None
    if not (_stack_eligible(module) and module.n_layers >= 2 and module.spec[1] <= 256):
        return None
    B, m, ns = idx.shape
    N = xyz.shape[1]
    Cf = feature.shape[-1]
    key = (xyz.device, B * m, ns)
    full = _FULL.get(key)
    if full is None:
        full = _FULL[key] = (torch.full((B, m), ns, dtype=torch.int32, device=xyz.device),
                             (torch.arange(B * m + 1, dtype=torch.int64, device=xyz.device) * ns).to(torch.int32))
    cnt, group_off = full
    need_dx = feature.requires_grad
    plan = _stack_plan(module, B * m * ns, 3 + Cf, ns, True, (B, N, m, Cf, 1), need_dx, 0, defer=True)
    ps, running = _stack_params(module)
    link = DeferLink()
    Y = _StackFn.apply(feature.contiguous(), (plan, (xyz.contiguous(), new_xyz.contiguous(), idx.contiguous(), cnt, group_off), running, link), *ps)
    return Y, link


def stack_grouped(module, xyz, new_xyz, feature, idx, cnt, group_off, use_xyz):
    B, m, ns = idx.shape
    N = xyz.shape[1]
    Cf = 0 if feature is None else feature.shape[-1]
    need_dx = feature is not None and feature.requires_grad
    c0 = (3 if use_xyz else 0) + Cf
    plan = _stack_plan(module, B * m * ns, c0, ns, True, (B, N, m, Cf, int(bool(use_xyz))), need_dx, 0)
    ps, running = _stack_params(module)
    feat = None if feature is None else feature.contiguous()
    out = _StackFn.apply(feat, (plan, (xyz, new_xyz, idx, cnt, group_off), running), *ps)
    return out.reshape(B, m, out.shape[-1])
