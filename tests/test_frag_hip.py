"""GPU: the fragment-direct GEMMs for layers with few rows (csrc/frag.hip) against fp64 PyTorch-CPU of the same op.

Forward (``pcl_frag_linear_fwd_f32``): Y and the BatchNorm sums, every launch shape (column pairs, K split 1 / 2 / 4), K tails (K = 259,
K % 8 = 4, K < 8), unaligned row strides, with and without the folded input BatchNorm, and the fp64-flushed accumulation, which must be
closer to fp64 than the plain fp32 chain.  Backward: ``pcl_frag_dy_f32`` (dense and sparse max gradient), ``pcl_frag_linear_bwd_dx_f32``
(masked epilogue + sums, first_col) and ``pcl_frag_linear_bwd_dw_f32`` (in-kernel reduction over the workgroups of a tile: bit-identical
from run to run).  Tolerances: 1e-5 * max(1, |ref|_max) on features (north_star), stated per check otherwise.
Reference: nn.Conv(k=1) + nn.BatchNorm + ReLU of networks/cls/pointnet2.py:25-29, :131-136 and misc/ops.py:54-64.
"""
import ctypes

import numpy as np
import pytest
import torch

from pointcloudlib_amd import _lib

pytestmark = pytest.mark.gpu


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def lrelu(t, s):
    return torch.maximum(t, t * s)


def fwd(dev, X, W, bias, sc, sh, slope, flush=0, stats=True, ldx=None, ldw=None, ldy=None):
    P, K = X.shape
    N = W.shape[0]
    ldx, ldw, ldy = ldx or K, ldw or K, ldy or N
    Xd = torch.zeros((P, ldx), device=dev); Xd[:, :K] = X.to(dev)
    Wd = torch.full((N, ldw), 7.0, device=dev); Wd[:, :K] = W.to(dev)
    if ldx > K:
        Xd[:, K:] = 3.0           # what lies between the rows must not matter
    Y = torch.full((P, ldy), -77.0, device=dev)
    rows = _lib.size_query("pcl_frag_stat_rows", P)
    st = torch.zeros((rows, 2, N), dtype=torch.float64, device=dev) if stats else None
    bd, scd, shd = (None if t is None else t.to(dev) for t in (bias, sc, sh))        # (kept alive across the call)
    _lib.call("pcl_frag_linear_fwd_f32", _p(Xd), ldx, _p(Wd), ldw, _p(bd), _p(scd), _p(shd), float(slope), P, K, N, _p(Y), ldy, None, _p(st), flush, _st())
    torch.cuda.synchronize()
    return Y.cpu(), (None if st is None else st.sum(0).cpu())


def ref_fwd(X, W, bias, sc, sh, slope):
    Xd = X.double()
    if sc is not None:
        Xd = lrelu(sc.double() * Xd + sh.double(), slope)
    Y = Xd @ W.double().t()
    if bias is not None:
        Y = Y + bias.double()
    return Y


FWD_CASES = [
    # P, K, N, act, bias, (force_tn, force_ksw)
    (4096, 259, 256, False, False, (0, 0)),      # GroupAll level, first layer: K tail of 3, unaligned rows
    (4096, 256, 512, True, False, (0, 0)),
    (2048, 512, 1024, True, False, (0, 0)),
    (2048, 1664, 256, False, True, (0, 0)),      # fp3 of the MSG decoder: conv bias, K = 1664
    (1000, 132, 96, True, True, (1, 2)),         # ragged rows (1000 = 15.6 row tiles), K % 8 = 4, N ends inside a column tile
    (777, 5, 40, False, True, (1, 1)),           # K < 8: only the partial step
    (640, 64, 200, True, False, (2, 1)),         # column pairs with N = 200: second pair half empty
    (640, 64, 200, True, False, (2, 4)),         # K split over four wave groups, 8 steps in all
    (130, 1024, 64, True, False, (1, 4)),
    (8192, 576, 256, False, False, (0, 0)),      # fp2 of the MSG decoder
]


@pytest.mark.parametrize("P,K,N,act,use_bias,force", FWD_CASES)
def test_frag_forward_matches_fp64(dev, P, K, N, act, use_bias, force):
    g = torch.Generator().manual_seed(P * 31 + K)
    X = torch.randn(P, K, generator=g) * 2.0 + 0.5
    W = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g) if use_bias else None
    sc = (torch.rand(K, generator=g) + 0.5) if act else None
    sh = torch.randn(K, generator=g) * 0.3 if act else None
    slope = 0.2 if (P % 2) else 0.0
    want = ref_fwd(X, W, bias, sc, sh, slope)
    _lib.lib().pcl_frag_set_tuning(-1, force[0], force[1], 0, 0, 0, 0)
    try:
        for ldx, ldw, ldy in ((None, None, None), (K + 3, K + 1, N + 5)):
            errs = {}
            for flush in (0, 8, 32):
                Y, st = fwd(dev, X, W, bias, sc, sh, slope, flush=flush, ldx=ldx, ldw=ldw, ldy=ldy)
                got = Y[:, :N].double()
                tol = 1e-5 * max(1.0, want.abs().max().item()) * (1 if flush else max(1.0, (K / 256) ** 0.5))
                err = (got - want).abs().max().item()
                assert err <= tol, f"flush {flush} ld {ldx}: max err {err:.3e} > {tol:.3e}"
                if ldy:
                    assert (Y[:, N:] == -77.0).all(), "columns past N were written"
                # BatchNorm sums of the STORED values
                s_ref, q_ref = got.sum(0), (got * got).sum(0)
                assert (st[0] - s_ref).abs().max().item() <= 1e-9 * max(1.0, s_ref.abs().max().item())
                assert (st[1] - q_ref).abs().max().item() <= 1e-9 * max(1.0, q_ref.abs().max().item())
                errs[flush] = (got - want).abs().mean().item()
            if K >= 256:      # the flushed accumulation is the more accurate one (mean error; fp32 storage rounding is the floor of both)
                assert errs[32] <= errs[0] and errs[8] <= errs[0], errs
            # no statistics: same values
            Y2, _ = fwd(dev, X, W, bias, sc, sh, slope, flush=0, stats=False, ldx=ldx, ldw=ldw, ldy=ldy)
            Y1, _ = fwd(dev, X, W, bias, sc, sh, slope, flush=0, stats=True, ldx=ldx, ldw=ldw, ldy=ldy)
            assert torch.equal(Y1, Y2)
    finally:
        _lib.lib().pcl_frag_set_tuning(-1, 0, 0, 0, 0, 0, 0)


def test_frag_forward_flush_gain(dev):
    """K = 1664 (the MSG decoder's fp3): the mean error against fp64 of the flushed accumulation, relative to the plain fp32 chain."""
    g = torch.Generator().manual_seed(5)
    P, K, N = 2048, 1664, 256
    X = torch.relu(torch.randn(P, K, generator=g))
    W = torch.randn(N, K, generator=g) / K ** 0.5
    want = ref_fwd(X, W, None, None, None, 0.0)
    e = {}
    for flush in (0, 32, 8):
        Y, _ = fwd(dev, X, W, None, None, None, 0.0, flush=flush)
        e[flush] = (Y.double() - want).abs().mean().item()
    floor = (want.float().double() - want).abs().mean().item()
    print(f"\n[frag forward K=1664] mean |err| vs fp64: fp32 chain {e[0]:.3e}, flush 32 {e[32]:.3e}, flush 8 {e[8]:.3e}; fp32 rounding of the exact value {floor:.3e}")
    assert e[32] < 0.5 * e[0] and e[8] <= 1.05 * e[32] and e[8] < 2.5 * floor


@pytest.mark.parametrize("sparse", [False, True])
def test_frag_dy(dev, sparse):
    g = torch.Generator().manual_seed(11)
    P, C, ns = 1536, 320, 128
    Y = torch.randn(P, C, generator=g)
    a, k1, k2, mu = (torch.randn(C, generator=g) for _ in range(4))
    G = P // ns
    if sparse:
        arg = torch.randint(0, ns, (G, C), generator=g, dtype=torch.int32)
        gz = torch.randn(G, C, generator=g)
        du = torch.zeros(P, C)
        rows = (torch.arange(G)[:, None] * ns + arg.long())
        du[rows, torch.arange(C)[None, :].expand(G, C)] = gz
    else:
        du = torch.randn(P, C, generator=g)
    want = a.double() * du.double() - (k1.double() + k2.double() * (Y.double() - mu.double()))
    dy = torch.empty(P, C, device=dev)
    zw = torch.full((37,), 9, dtype=torch.int32, device=dev)
    d = lambda t: t.to(dev)
    keep = [d(Y), d(a), d(k1), d(k2), d(mu)]
    if sparse:
        keep += [d(arg), d(gz)]
        _lib.call("pcl_frag_dy_f32", None, _p(keep[0]), _p(keep[1]), _p(keep[2]), _p(keep[3]), _p(keep[4]), _p(keep[5]), _p(keep[6]), ns, P, C, _p(dy),
                  _p(zw), 37, _st())
    else:
        keep += [d(du)]
        _lib.call("pcl_frag_dy_f32", _p(keep[5]), _p(keep[0]), _p(keep[1]), _p(keep[2]), _p(keep[3]), _p(keep[4]), None, None, 1, P, C, _p(dy), _p(zw), 37,
                  _st())
    torch.cuda.synchronize()
    assert (dy.cpu().double() - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())
    assert (zw == 0).all()


DX_CASES = [
    # P, Cout, Cin, masked, first_col, force
    (4096, 256, 259, False, 3, (0, 0)),       # the GroupAll level's input gradient: plain, xyz columns skipped, unaligned rows
    (4096, 512, 256, True, 0, (0, 0)),
    (2048, 1024, 512, True, 0, (0, 0)),
    (1000, 96, 132, True, 0, (1, 2)),
    (333, 64, 40, True, 0, (2, 1)),
    (640, 200, 64, False, 0, (1, 4)),
]


@pytest.mark.parametrize("P,Cout,Cin,masked,first_col,force", DX_CASES)
def test_frag_dx_matches_fp64(dev, P, Cout, Cin, masked, first_col, force):
    g = torch.Generator().manual_seed(P + Cout)
    dy = torch.randn(P, Cout, generator=g)
    W = torch.randn(Cout, Cin, generator=g) / Cout ** 0.5
    Yp = torch.randn(P, Cin, generator=g)
    psc, psh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    slope = 0.2 if P % 2 else 0.0
    want = dy.double() @ W.double()
    if masked:
        pre = psc.double() * Yp.double() + psh.double()
        want = torch.where(pre > 0, want, want * slope)
    d = lambda t: t.to(dev)
    dyd, Wd, Ypd, pscd, pshd = d(dy), d(W), d(Yp), d(psc), d(psh)
    out = torch.full((P, Cin), -5.0, device=dev)
    rows = _lib.size_query("pcl_frag_stat_rows", P)
    st = torch.zeros((rows, 2, Cin), dtype=torch.float64, device=dev)
    _lib.lib().pcl_frag_set_tuning(-1, force[0], force[1], 0, 0, 0, 0)
    try:
        _lib.call("pcl_frag_linear_bwd_dx_f32", _p(dyd), _p(Wd), Cin, P, Cout, Cin, _p(Ypd) if masked else None, Cin, _p(pscd) if masked else None,
                  _p(pshd) if masked else None, float(slope), _p(out), Cin, _p(st) if masked else None, first_col, _st())
    finally:
        _lib.lib().pcl_frag_set_tuning(-1, 0, 0, 0, 0, 0, 0)
    torch.cuda.synchronize()
    got = out.cpu().double()
    # a pre-activation within rounding of zero may take the other branch of the mask: compare where |pre| is clear of zero
    ok = torch.ones_like(want, dtype=torch.bool)
    if masked:
        ok = (psc.double() * Yp.double() + psh.double()).abs() > 1e-5
    tol = 1e-5 * max(1.0, want.abs().max().item()) * max(1.0, (Cout / 256) ** 0.5)
    assert ((got - want).abs() * ok)[:, first_col:].max().item() <= tol
    if first_col:
        assert (out[:, :first_col] == -5.0).all(), "columns below first_col were written"
    if masked:
        s = st.sum(0).cpu()
        assert (s[0] - got.sum(0)).abs().max().item() <= 1e-9 * max(1.0, got.abs().sum(0).max().item())
        assert (s[1] - (got * Yp.double()).sum(0)).abs().max().item() <= 1e-9 * max(1.0, (got * Yp.double()).abs().sum(0).max().item())


DW_CASES = [
    # P, Cout, Cin, act, (tm, tn, ksw, ksg)
    (4096, 256, 259, False, (0, 0, 0, 0)),
    (4096, 512, 256, True, (0, 0, 0, 0)),
    (4096, 1024, 512, True, (0, 0, 0, 0)),
    (2048, 256, 1664, False, (0, 0, 0, 0)),
    (1000, 96, 132, True, (1, 1, 2, 3)),
    (1000, 96, 132, True, (2, 2, 1, 1)),
    (1000, 200, 70, True, (1, 2, 4, 2)),
    (1000, 200, 70, False, (2, 1, 4, 5)),
    (50, 64, 64, True, (1, 1, 4, 4)),          # fewer rows than row groups: empty wave groups and empty workgroups
]


@pytest.mark.parametrize("P,Cout,Cin,act,shape", DW_CASES)
def test_frag_dw_matches_fp64_and_is_deterministic(dev, P, Cout, Cin, act, shape):
    g = torch.Generator().manual_seed(P + Cin)
    dy = torch.randn(P, Cout, generator=g)
    X = torch.randn(P, Cin, generator=g)
    psc, psh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    slope = 0.2 if Cin % 2 else 0.0
    z = lrelu(psc.double() * X.double() + psh.double(), slope) if act else X.double()
    want = dy.double().t() @ z
    d = lambda t: t.to(dev)
    dyd, Xd, pscd, pshd = d(dy), d(X), d(psc), d(psh)
    ldo = Cin + 2
    L = _lib.lib()
    L.pcl_frag_set_tuning(-1, 0, 0, *shape)
    try:
        nbytes = L.pcl_frag_dw_workspace_bytes(P, Cout, Cin)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev).fill_(0x5a)
        outs = []
        for rep in range(4):
            out = torch.full((Cout, ldo), -3.0, device=dev)
            _lib.call("pcl_frag_linear_bwd_dw_f32", _p(dyd), _p(Xd), Cin, _p(pscd) if act else None, _p(pshd) if act else None, float(slope), P, Cout, Cin,
                      _p(out), ldo, _p(ws), nbytes, 0 if rep == 0 else 1, _st())      # (later runs: the last arrivers re-zeroed their counters)
            torch.cuda.synchronize()
            outs.append(out.cpu())
    finally:
        L.pcl_frag_set_tuning(-1, 0, 0, 0, 0, 0, 0)
    got = outs[0][:, :Cin].double()
    scale = (dy.double().abs().t() @ z.abs()).max().item()
    assert (got - want).abs().max().item() <= 2e-6 * scale, f"{(got - want).abs().max().item():.3e} vs {2e-6 * scale:.3e}"
    assert (outs[0][:, Cin:] == -3.0).all()
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), "the in-kernel reduction is not deterministic"
