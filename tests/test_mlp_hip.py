"""GPU: the fused MFMA MLP path (csrc/mlp.hip) against a plain PyTorch reference of the same op
(fp64 on CPU as ground truth; fp32 tolerance stated per check).  Features: |err| <= 1e-5 * max(1, |ref|_max)
(north_star: "within 1e-5 fp32 for features"); gradients: 1e-4 relative to the gradient's max-norm, and never
worse than 4x the error of PyTorch's own fp32 GPU path against the same fp64 truth."""
import copy

import numpy as np

import pytest
import torch

import oracle.torch_backend  # noqa: F401,E402  (registers the plain-PyTorch composite the tests compare against)

from pointcloudlib_amd.misc.layers import PointwiseMLP

pytestmark = pytest.mark.gpu


def run(module, x, ns, gout, backend):
    module.backend = backend
    module.zero_grad()
    x = x.clone().requires_grad_(True)
    out = module(x, group_max=ns)
    out.backward(gout)
    grads = {n: p.grad.detach().clone() for n, p in module.named_parameters()}
    bufs = {n: b.detach().clone() for n, b in module.named_buffers()}
    return out.detach(), x.grad.detach(), grads, bufs


CASES = [
    # spec, lead shape, group_max, bias, slope
    ([6, 64, 64, 128], (2, 25, 16), 16, False, 0.0),        # SA1-like, K=6 (scalar staging), ragged P=800
    ([131, 128, 128, 256], (2, 16, 32), 32, False, 0.0),    # SA2-like, K=131, N=256 (two column tiles)
    ([259, 256, 512, 1024], (3, 1, 128), 128, False, 0.0),  # SA3-like (GroupAll), deep K
    ([6, 64, 64, 128], (3, 5, 64), 64, False, 0.0),         # ns = 64: max/min fused into the last GEMM epilogue
    ([35, 48], (2, 9, 64), 64, True, 0.2),                  # fused max, single layer, bias, LeakyReLU, ragged tiles
    ([6, 64], (2, 40, 20), 20, False, 0.2),                 # DGCNN edge conv: LeakyReLU(0.2), max over k
    ([64, 40, 24], (3, 70), None, True, 0.0),               # FP-style: bias + BN + ReLU, no max, N < 128
    ([12, 8, 8, 16], (2, 33, 8), None, True, 0.0),          # WeightNet-sized
    ([10, 300], (2, 7, 8), 8, False, 0.0),                  # > 256 channels with a max: two channel blocks in the backward prep
    ([7, 5, 3], (4, 50), None, True, 0.2),                  # 3 output channels: 4 channel-lanes x 64 row-lanes per block
    ([64, 64, 128], (13, 8, 16), 16, False, 0.0),           # 1664 rows = 26 row tiles: full groups of 8 AND a remainder in the
                                                            # XCD-aware block -> tile mapping, two column tiles
    ([32, 192], (11, 3, 32), None, False, 0.0),             # 1056 rows, three column tiles, no max
    ([6, 64, 96, 128], (3, 40, 32), 32, False, 0.0),        # the MSG stacks' 64 -> 96 -> 128: N = 96 ends inside a column tile (per-block
                                                            # epilogue decision), forward AND dX (Cin = 96)
    ([20, 96, 32], (2, 300), None, True, 0.0),              # 96 wide without a max, bias
    ([16, 160, 48], (5, 64), None, False, 0.2),             # 160 = 128 + 32: a second column tile with one live 32-column block
]


# the fused backward (dX + dW of a layer in one persistent kernel): every (Cout, Cin) instantiation, dense and sparse
FB_CASES = [
    ([16, 128, 64, 256], (5, 25, 32), 32, False, 0.0),      # 4000 rows: 256x64 sparse, 64x128 dense
    ([16, 64, 128, 64], (3, 1111), None, False, 0.2),       # 3333 rows (ragged last tile), no max: 64x128 and 128x64 dense
    ([9, 128, 128, 256], (2, 30, 48), 48, True, 0.0),       # 2880 rows: 256x128 sparse, 128x128 dense
    ([5, 64, 64, 64, 128], (7, 9, 16), 16, False, 0.0),     # 1008 rows: 128x64 sparse, 64x64 dense twice
    ([4, 128, 128, 256], (2, 5, 4), 4, False, 0.0),         # 40 rows: one ragged tile, most workgroups of a capped grid idle
    ([4, 64, 128, 64], (1, 65), None, True, 0.2),           # 65 rows: one full 64-row tile + 1 row (128-row tiles: one ragged)
]


@pytest.mark.parametrize("spec,lead,ns,bias,slope", CASES)
def test_fused_mlp_matches_reference(dev, spec, lead, ns, bias, slope):
    _check_against_fp64(dev, spec, lead, ns, bias, slope)


@pytest.mark.parametrize("blocks", [3, 256])
@pytest.mark.parametrize("spec,lead,ns,bias,slope", FB_CASES)
def test_fused_backward_tiles_per_block(dev, monkeypatch, spec, lead, ns, bias, slope, blocks):
    """``blocks`` = 3: every workgroup of the persistent kernel walks many row tiles (prefetch of the next tile's operands,
    the cyclic weight-chunk stage, the double-buffered row records), at a row count where an fp64 comparison is still free
    of ReLU-mask flips (a pre-activation within fp32 rounding of 0 turns up about once per 10^7 elements)."""
    from pointcloudlib_amd import _lib
    from pointcloudlib_amd.misc import mlp_hip
    _lib.lib().pcl_set_fb_max_blocks(blocks)
    _lib.size_query.cache_clear(); mlp_hip._PLANS.clear()          # sizes depend on the cap
    try:
        _check_against_fp64(dev, spec, lead, ns, bias, slope)
    finally:
        _lib.lib().pcl_set_fb_max_blocks(0)
        _lib.size_query.cache_clear(); mlp_hip._PLANS.clear()


@pytest.mark.parametrize("spec,lead,ns", [([16, 128, 64, 256], (8, 512, 32), 32), ([16, 64, 128, 64], (3, 44444), None),
                                          ([9, 128, 128, 256], (4, 500, 48), 48)])
def test_fused_backward_equals_split_kernels_at_size(dev, spec, lead, ns):
    """Full-size rows (2-9 tiles per workgroup): the fused kernel against the separate dX and dW kernels.  Both take the
    ReLU masks from the same stored pre-activations with the same arithmetic, so (unlike an fp64 reference at 10^5 rows)
    no mask can differ and the two agree to summation-order rounding."""
    from pointcloudlib_amd.misc import mlp_hip
    torch.manual_seed(99)
    m = PointwiseMLP(spec, bias=False, slope=0.0).to(dev)
    x = torch.randn(*lead, spec[0], device=dev)
    gout = torch.randn((*lead[:-1], spec[-1]) if ns else (*lead, spec[-1]), device=dev)
    res = {}
    for fused in (True, False):
        old, mlp_hip._FUSED_BWD = mlp_hip._FUSED_BWD, fused
        try:
            res[fused] = run(copy.deepcopy(m), x, ns, gout, "hip")
        finally:
            mlp_hip._FUSED_BWD = old
    f, s = res[True], res[False]
    assert torch.equal(f[0], s[0])
    for name, a, b in [("x", f[1], s[1])] + [(n, f[2][n], s[2][n]) for n in s[2]]:
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() <= 2e-5 * max(scale, 1e-6) + 1e-7, (name, (a - b).abs().max().item(), scale)


def _check_against_fp64(dev, spec, lead, ns, bias, slope):
    torch.manual_seed(1234 + spec[0])
    m64 = PointwiseMLP(spec, bias=bias, slope=slope).double()
    with torch.no_grad():
        for g, b in zip(m64.gammas, m64.betas):
            g.uniform_(0.5, 1.5); b.uniform_(-0.3, 0.3)
        for g in m64.gammas:
            g[::3] *= -1.0                                   # negative gamma: max must not assume monotone BN
    x64 = torch.randn(*lead, spec[0], dtype=torch.float64)
    if ns:                                                   # padded duplicates inside groups (ball-query padding)
        x64[..., ns // 2:, :] = x64[..., :1, :]
    m64.backend = "torch"
    out_shape = (*lead[:-1], spec[-1]) if ns else (*lead, spec[-1])
    gout64 = torch.randn(out_shape, dtype=torch.float64)
    ref = run(copy.deepcopy(m64), x64, ns, gout64, "torch")

    m32 = copy.deepcopy(m64).float().to(dev)
    x32, g32 = x64.float().to(dev), gout64.float().to(dev)
    t32 = run(copy.deepcopy(m32), x32, ns, g32, "torch")
    h32 = run(copy.deepcopy(m32), x32, ns, g32, "hip")

    def err(a, b):
        return (a.double().cpu() - b).abs().max().item()

    scale = max(1.0, ref[0].abs().max().item())
    e_h, e_t = err(h32[0], ref[0]), err(t32[0], ref[0])
    assert e_h <= 1e-5 * scale, f"features: hip err {e_h:.3e} (torch fp32 {e_t:.3e}), scale {scale:.3f}"
    gs = max(1e-6, ref[1].abs().max().item())
    e_h, e_t = err(h32[1], ref[1]), err(t32[1], ref[1])
    assert e_h <= max(1e-4 * gs, 4 * e_t), f"input grad: hip {e_h:.3e} torch {e_t:.3e} scale {gs:.3e}"
    for name in ref[2]:
        if "biases" in name:                                 # bias feeds BatchNorm: true gradient is 0
            assert h32[2][name].abs().max().item() <= 1e-4
            continue
        gs = max(1e-6, ref[2][name].abs().max().item())
        e_h, e_t = err(h32[2][name], ref[2][name]), err(t32[2][name], ref[2][name])
        assert e_h <= max(1e-4 * gs, 4 * e_t), f"{name}: hip {e_h:.3e} torch {e_t:.3e} scale {gs:.3e}"
    for name in ref[3]:                                      # running statistics (Jittor rule: biased variance)
        assert err(h32[3][name], ref[3][name]) <= 1e-5 * max(1.0, ref[3][name].abs().max().item()), name


def test_fused_mlp_eval_mode_and_no_input_grad(dev):
    torch.manual_seed(0)
    m = PointwiseMLP([6, 32, 64]).to(dev)
    x = torch.randn(4, 10, 8, 6, device=dev)
    m.train(); m.backend = "hip"; m(x, group_max=8)          # populate running stats
    m.eval()
    m.backend = "torch"; a = m(x, group_max=8)
    m.backend = "hip"; b = m(x, group_max=8)
    assert (a - b).abs().max().item() <= 1e-5 * max(1.0, a.abs().max().item())
    m.train()
    out = m(x, group_max=8)                                  # x does not require grad: no dX GEMM for layer 0
    out.sum().backward()
    assert all(p.grad is not None for p in m.parameters())


def test_full_size_sa1_statistics_property(dev):
    """BASELINE size (P = 32*512*64 rows): after BN the pre-activation has mean 0 / var 1 per channel, so the
    reported batch statistics must reproduce those of an fp64 reduction of the stored pre-BN output."""
    torch.manual_seed(3)
    m = PointwiseMLP([6, 64, 64, 128]).to(dev).train()
    m.backend = "hip"
    x = torch.randn(32, 512, 64, 6, device=dev)
    m(x, group_max=64)
    y = torch.nn.functional.linear(x.reshape(-1, 6), m.weights[0])
    mean = y.double().mean(0)
    var = y.double().var(0, unbiased=False)
    rm, rv = m.running_mean_0.double(), m.running_var_0.double()
    assert (rm - 0.1 * mean).abs().max().item() < 1e-6
    assert (rv - (0.9 + 0.1 * var)).abs().max().item() < 1e-5


@pytest.mark.parametrize("last", [64, 30])      # 30: Cout % 4 != 0 -> the scalar staging of the sparse max gradient on compacted rows (64-row tiles)
def test_duplicate_compacted_rows_match_padded_groups(oracle, dev, last):
    """The ragged path (MLP on the DISTINCT rows of ball-query groups + multiplicities) must reproduce the padded
    computation: pooled features, every parameter gradient, the feature gradient and the running statistics."""
    from pointcloudlib_amd import synth
    from pointcloudlib_amd.misc import ops
    torch.manual_seed(11)
    B, N, m, ns, C = 4, 512, 96, 32, 13
    pts = synth.gauss_ball(B, N, 31)
    x = torch.from_numpy(pts).to(dev)
    feat0 = torch.randn(B, N, C, device=dev)
    idx_f, new_xyz = ops.furthest_point_sample(x, m)
    idx, cnt = ops.ball_query(new_xyz, x, 0.25, ns, return_cnt=True)
    assert 0.2 < (cnt.float().mean().item() / ns) < 0.95          # a real mix of padded and saturated groups
    mlp = PointwiseMLP([3 + C, 32, 32, last]).to(dev).train()
    with torch.no_grad():
        for g in mlp.gammas:
            g.uniform_(0.5, 1.5); g[::4] *= -1.0
    gout = torch.randn(B, m, last, device=dev)
    res = []
    for compact in (False, True):
        mm = copy.deepcopy(mlp)
        f = feat0.clone().requires_grad_(True)
        if compact:
            rows, rs = ops.group_points_compact(x, new_xyz, f, idx, cnt, True)
            out = mm(rows, rowset=rs)
            n_rows = int(rs.n_rows_dev.item())
            assert n_rows == int(cnt.clamp(min=1).sum().item())
        else:
            out = mm(ops.group_points(x, new_xyz, f, idx, True), group_max=ns)
        out.backward(gout)
        res.append((out.detach(), f.grad.detach(), {n: p.grad.detach() for n, p in mm.named_parameters()},
                    {n: b.detach().clone() for n, b in mm.named_buffers()}))
    (o0, f0, g0, b0), (o1, f1, g1, b1) = res
    assert (o0 - o1).abs().max().item() <= 1e-5 * max(1.0, o0.abs().max().item())
    assert (f0 - f1).abs().max().item() <= 1e-4 * max(1e-6, f0.abs().max().item())
    for n in g0:
        assert (g0[n] - g1[n]).abs().max().item() <= 1e-4 * max(1e-6, g0[n].abs().max().item()), n
    for n in b0:
        assert (b0[n] - b1[n]).abs().max().item() <= 1e-5 * max(1.0, b0[n].abs().max().item()), n


@pytest.mark.parametrize("radius,ns", [(1e-4, 16), (10.0, 8), (0.3, 1), (0.25, 64)])
def test_compacted_rows_edge_cases(dev, radius, ns):
    """Groups with a single hit (all padding), saturated groups (no padding), ns = 1, and queries without any hit
    (defined as ns copies of point 0): the compacted path equals the padded path."""
    from pointcloudlib_amd import synth
    from pointcloudlib_amd.misc import ops
    torch.manual_seed(5)
    B, N, m, C = 3, 256, 40, 5
    x = torch.from_numpy(synth.gauss_ball(B, N, 41)).to(dev)
    feat0 = torch.randn(B, N, C, device=dev)
    _, new_xyz = ops.furthest_point_sample(x, m)
    new_xyz = new_xyz.clone()
    new_xyz[:, -3:] += 50.0                                      # three queries per cloud hit nothing
    idx, cnt = ops.ball_query(new_xyz, x, radius, ns, return_cnt=True)
    assert (cnt[:, -3:] == 0).all()
    mlp = PointwiseMLP([3 + C, 16, 24]).to(dev).train()
    gout = torch.randn(B, m, 24, device=dev)
    outs = []
    for compact in (False, True):
        mm = copy.deepcopy(mlp)
        f = feat0.clone().requires_grad_(True)
        if compact:
            rows, rs = ops.group_points_compact(x, new_xyz, f, idx, cnt, True)
            assert int(rs.n_rows_dev.item()) == int(cnt.clamp(min=1).sum().item())
            out = mm(rows, rowset=rs)
        else:
            out = mm(ops.group_points(x, new_xyz, f, idx, True), group_max=ns)
        out.backward(gout)
        outs.append((out.detach(), f.grad.detach(), [p.grad.detach() for p in mm.parameters()]))
    (o0, f0, g0), (o1, f1, g1) = outs
    assert (o0 - o1).abs().max().item() <= 1e-5 * max(1.0, o0.abs().max().item())
    assert (f0 - f1).abs().max().item() <= 1e-4 * max(1e-6, f0.abs().max().item())
    for a, b in zip(g0, g1):
        assert (a - b).abs().max().item() <= 1e-4 * max(1e-6, a.abs().max().item())


@pytest.mark.parametrize("offset,spread", [(5.0, 0.01), (0.3, 1e-3), (10.0, 1.0)])
def test_batchnorm_statistics_survive_large_mean(dev, offset, spread):
    """|mean| >> std (PointConv's DensityNet sees near-constant densities): the GEMM epilogue sums about a pivot, so
    the batch statistics must be as good as PyTorch's own fp32 BatchNorm against an fp64 ground truth.
    (BatchNorm is APPLIED as one fused multiply-add y*scale+shift, whose rounding is ~6e-8*|mean|/std of a normalised
    unit; at |mean|/std in the hundreds a ReLU mask can flip for values within that distance of zero.)"""
    torch.manual_seed(0)
    m = PointwiseMLP([8, 16, 16], bias=True).to(dev).train()
    x = offset + spread * torch.randn(4096, 8, device=dev)
    gout = torch.randn(4096, 16, device=dev)
    o_h, gx_h, g_h, _ = run(copy.deepcopy(m), x, None, gout, "hip")
    o_t, gx_t, g_t, _ = run(copy.deepcopy(m), x, None, gout, "torch")
    o_d, gx_d, g_d, _ = run(copy.deepcopy(m).double(), x.double(), None, gout.double(), "torch")
    scale = max(1.0, o_d.abs().max().item())
    e_h, e_t = (o_h.double() - o_d).abs().max().item(), (o_t.double() - o_d).abs().max().item()
    assert e_h <= max(4 * e_t, 1e-5 * scale), (e_h, e_t)
    for n in g_d:
        s = max(1e-6, g_d[n].abs().max().item())
        eh, et = (g_h[n].double() - g_d[n]).abs().max().item(), (g_t[n].double() - g_d[n]).abs().max().item()
        assert eh <= max(4 * et, 1e-4 * s), (n, eh, et)


@pytest.mark.parametrize("R,K,N,bn,slope,bias", [(32, 1024, 512, True, 0.0, False), (32, 256, 40, False, 1.0, True),
                                                 (5, 300, 33, True, 0.2, True), (64, 2048, 512, True, 0.2, False),
                                                 (1, 16, 8, False, 0.0, True), (33, 70, 130, True, 0.0, True)])
def test_head_layer_matches_torch(dev, R, K, N, bn, slope, bias):
    """csrc/head.hip (Linear + BatchNorm1d + activation on <= 64 rows as one kernel) against the PyTorch modules in fp64:
    outputs, running statistics and every gradient; training and evaluation mode."""
    from torch import nn
    from pointcloudlib_amd.misc.head import head_layer
    torch.manual_seed(R + K)
    lin = nn.Linear(K, N, bias=bias).to(dev)
    b = nn.BatchNorm1d(N).to(dev) if bn else None
    if bn:
        b.weight.data.uniform_(-1.0, 1.5); b.bias.data.uniform_(-0.5, 0.5)
    act = None if slope == 1.0 else (nn.ReLU() if slope == 0.0 else nn.LeakyReLU(slope))
    for training in ((True, False) if R > 1 else (False,)):
        lin_d, b_d = copy.deepcopy(lin).double(), (copy.deepcopy(b).double() if bn else None)
        for m in (b, b_d):
            if m is not None:
                m.train(training)
        x = torch.randn(R, K, device=dev, requires_grad=True)
        g = torch.randn(R, N, device=dev)
        rv0 = b_d.running_var.clone() if bn else None
        out = head_layer(x, lin, b, act)
        out.backward(g)
        xd = x.detach().double().requires_grad_(True)
        y = lin_d(xd)
        if bn:
            y = b_d(y)
        ref = y if act is None else torch.nn.functional.leaky_relu(y, slope)
        ref.backward(g.double())
        pairs = [("out", out.detach(), ref.detach()), ("dx", x.grad, xd.grad), ("dW", lin.weight.grad, lin_d.weight.grad)]
        if bias:
            pairs.append(("db", lin.bias.grad, lin_d.bias.grad))
        if bn:
            # running_var follows Jittor's nn.BatchNorm: r += (biased batch variance - r) * momentum, where
            # torch's BatchNorm1d uses the unbiased one -- redo the reference's update with the biased variance
            rvar_ref = b_d.running_var
            if training:
                rvar_ref = rv0 + (lin_d(xd).detach().var(dim=0, unbiased=False) - rv0) * b_d.momentum
            pairs += [("dgamma", b.weight.grad, b_d.weight.grad), ("dbeta", b.bias.grad, b_d.bias.grad),
                      ("rmean", b.running_mean, b_d.running_mean), ("rvar", b.running_var, rvar_ref)]
        for name, a, r in pairs:
            s = max(1.0, r.abs().max().item())
            assert (a.double() - r).abs().max().item() <= 2e-5 * s, (name, training)
        for m in (lin, b):
            if m is not None:
                m.zero_grad()


@pytest.mark.gpu
@pytest.mark.parametrize("R,C", [(32, 40), (1, 40), (7, 2), (300, 130), (64, 1000)])
def test_soft_ce_kernel_matches_composite(R, C):
    """pcl_soft_ce_f32 (loss + gradient, one launch) vs the composite restatement of train_cls.py:31-51 in fp64."""
    from pointcloudlib_amd.train_utils import soft_cross_entropy_loss
    torch.manual_seed(R * 131 + C)
    x = (torch.randn(R, C, device="cuda") * 4).requires_grad_(True)
    t = torch.randint(0, C, (R,), device="cuda")
    loss = soft_cross_entropy_loss(x, t)
    (loss * 1.7).backward()
    xd = x.detach().double().cpu().requires_grad_(True)
    ref = soft_cross_entropy_loss(xd, t.cpu())            # CPU tensors take the composite path
    (ref * 1.7).backward()
    assert abs(loss.item() - ref.item()) <= 2e-6 * max(1.0, abs(ref.item()))
    assert torch.allclose(x.grad.cpu().double(), xd.grad, rtol=1e-5, atol=1e-7)
    with torch.no_grad():                                  # no gradient requested: forward only
        assert abs(float(soft_cross_entropy_loss(x.detach(), t)) - float(ref)) <= 2e-6 * max(1.0, abs(ref.item()))


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,C", [(16, 2048, 50), (1, 1, 50), (3, 777, 7), (2, 40000, 130)])
def test_seg_cross_entropy_kernel_matches_torch(B, N, C):
    """pcl_soft_ce_rows_f32 (train_partseg.py:116: mean cross entropy over every point; loss + gradient from one kernel and a
    fixed-order fold) on the [B, C, N] VIEW the part-seg networks return, against F.cross_entropy in fp64; run-to-run identical."""
    import torch.nn.functional as F
    from pointcloudlib_amd.train_utils import seg_cross_entropy_loss
    torch.manual_seed(B * 7 + C)
    rows = (torch.randn(B, N, C, device="cuda") * 3).requires_grad_(True)        # the head's layout
    seg = torch.randint(0, C, (B, N), device="cuda")
    loss = seg_cross_entropy_loss(rows.permute(0, 2, 1), seg)
    (loss * 0.6).backward()
    xd = rows.detach().double().cpu().requires_grad_(True)
    ref = F.cross_entropy(xd.permute(0, 2, 1), seg.cpu())
    (ref * 0.6).backward()
    assert abs(loss.item() - ref.item()) <= 3e-6 * max(1.0, abs(ref.item()))
    assert torch.allclose(rows.grad.cpu().double(), xd.grad, rtol=1e-5, atol=1e-9)
    again = seg_cross_entropy_loss(rows.detach().permute(0, 2, 1), seg)
    assert again.item() == loss.item()
    flat = seg_cross_entropy_loss(rows.detach().reshape(-1, C), seg.reshape(-1))             # the [R, C] form train_partseg.py passes
    assert flat.item() == loss.item()


@pytest.mark.gpu
@pytest.mark.parametrize("R,K,N,bias", [(32, 16384, 1024, True), (7, 2048, 20, False), (32, 4096, 130, True)])
def test_wide_single_layer_takes_head_kernels(dev, R, K, N, bias):
    """PointwiseMLP([K, N]) on <= 32 rows with K >= 2048 (PointConv's per-point Linear on the GroupAll level) runs on the
    8-columns-per-workgroup head kernels: forward, all gradients and the (biased-variance) running statistics against the
    fp64 PyTorch restatement."""
    torch.manual_seed(R + N)
    m64 = PointwiseMLP([K, N], bias=bias).double()
    m = copy.deepcopy(m64).float().to(dev)
    m64.backend = "torch"
    x = torch.randn(2, R // 2 if R % 2 == 0 else R, K, dtype=torch.float64)[:1 if R % 2 else 2]
    x = x.reshape(-1, K)[:R].contiguous()
    g = torch.randn(R, N, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    want = m64(xr); want.backward(g)
    xg = x.float().to(dev).requires_grad_(True)
    got = m(xg); got.backward(g.float().to(dev))
    tol = lambda t: 3e-5 * max(1.0, t.abs().max().item())
    assert (got.double().cpu() - want).abs().max().item() <= tol(want)
    assert (xg.grad.double().cpu() - xr.grad).abs().max().item() <= tol(xr.grad)
    for (n, p), (_, q) in zip(m.named_parameters(), m64.named_parameters()):
        assert (p.grad.double().cpu() - q.grad).abs().max().item() <= tol(q.grad), n
    for b in ("running_mean_0", "running_var_0"):
        assert torch.allclose(getattr(m, b).double().cpu(), getattr(m64, b), rtol=1e-5, atol=1e-6), b


# ---- per-stack entry points (csrc/stack.hip): one C call per stack and direction, the same kernels in the same order ----
def _with_stack(flag, fn):
    from pointcloudlib_amd.misc import mlp_hip
    old = mlp_hip.USE_STACK
    mlp_hip.USE_STACK = flag
    try:
        return fn()
    finally:
        mlp_hip.USE_STACK = old


@pytest.mark.parametrize("spec,lead,ns,bias,slope", CASES + FB_CASES)
def test_stack_entry_points_equal_per_kernel_path_plain(dev, spec, lead, ns, bias, slope):
    """pcl_mlp_stack_fwd/bwd_f32 launch the kernels the per-kernel path launches, in the same order, on the same operands:
    outputs, input gradient, every parameter gradient and the running statistics must be BIT-identical."""
    torch.manual_seed(3)
    mlp = PointwiseMLP(spec, bias=bias, slope=slope).to(dev).train()
    with torch.no_grad():
        for g in mlp.gammas:
            g.uniform_(0.5, 1.5); g[::3] *= -1.0
    x = torch.randn(*lead, spec[0], device=dev)
    gshape = (lead[:-1] if ns else lead) + (spec[-1],)
    gout = torch.randn(*gshape, device=dev)
    a = _with_stack(False, lambda: run(copy.deepcopy(mlp), x, ns, gout, "auto"))
    b = _with_stack(True, lambda: run(copy.deepcopy(mlp), x, ns, gout, "auto"))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for n in a[2]:
        assert torch.equal(a[2][n], b[2][n]), n
    for n in a[3]:
        assert torch.equal(a[3][n], b[3][n]), n


@pytest.mark.parametrize("C,feat_grad,use_xyz,spec_tail", [(3, False, True, [64, 64, 128]), (13, True, True, [32, 32, 64]),
                                                          (128, True, True, [128, 128, 256]), (3, True, True, [64, 128]),
                                                          (16, False, False, [64, 64]), (0, False, True, [32, 64, 64])])
def test_stack_entry_points_equal_per_kernel_path_grouped(dev, C, feat_grad, use_xyz, spec_tail):
    """The grouped stack (ball-query grouping folded into the first layer, duplicate-compacted rows, max over the group) as
    one call forward and one backward against the two-node per-kernel path: identical forward; gradients identical where
    no atomics are involved (everything but the scatter to the points), else to fp32 summation order."""
    from pointcloudlib_amd import synth
    from pointcloudlib_amd.misc import ops
    torch.manual_seed(7)
    B, N, m, ns = 4, 512, 96, 32
    x = torch.from_numpy(synth.gauss_ball(B, N, 31)).to(dev)
    feat0 = torch.randn(B, N, C, device=dev) if C else None
    _, new_xyz = ops.furthest_point_sample(x, m)
    idx, cnt = ops.ball_query(new_xyz, x, 0.25, ns, return_cnt=True)
    goff = ops.group_offsets(cnt)
    mlp = PointwiseMLP([(3 if use_xyz else 0) + C] + spec_tail).to(dev).train()
    with torch.no_grad():
        for g in mlp.gammas:
            g.uniform_(0.5, 1.5); g[::4] *= -1.0
    gout = torch.randn(B, m, spec_tail[-1], device=dev)

    def go():
        mm = copy.deepcopy(mlp)
        f = None if feat0 is None else feat0.clone().requires_grad_(feat_grad)
        out = mm.forward_grouped(x, new_xyz, f, idx, cnt, goff, use_xyz)
        out.backward(gout)
        return (out.detach(), None if (f is None or not feat_grad) else f.grad.detach(), {n: p.grad.detach() for n, p in mm.named_parameters()},
                {n: b.detach().clone() for n, b in mm.named_buffers()})

    a, b = _with_stack(False, go), _with_stack(True, go)
    assert torch.equal(a[0], b[0])
    for n in a[3]:
        assert torch.equal(a[3][n], b[3][n]), n
    atomics = C > 4 or feat_grad                      # the wide-feature path scatters dy to the points with fp32 atomics
    for n in a[2]:
        if atomics and n == "weights.0":
            assert (a[2][n] - b[2][n]).abs().max().item() <= 1e-5 * max(1e-6, a[2][n].abs().max().item()), n
        else:
            assert torch.equal(a[2][n], b[2][n]), n
    if a[1] is not None:
        assert (a[1] - b[1]).abs().max().item() <= 1e-5 * max(1e-6, a[1].abs().max().item())


@pytest.mark.parametrize("grouped", [True, False])
def test_stack_backward_reads_a_column_slice_of_a_wider_gradient_in_place(dev, grouped):
    """Multi-scale grouping concatenates the pooled outputs of several stacks: each stack's backward gets a column slice of the
    wide gradient (row stride = the concatenated width).  The max-gradient kernel reads it in place (pcl_mlp_stack_t.gout_ld):
    same bits as from a dense copy, and no copy made."""
    from pointcloudlib_amd import synth
    from pointcloudlib_amd.misc import ops, mlp_hip
    torch.manual_seed(11)
    B, N, m, ns, cl = 3, 512, 64, 32, 128
    x = torch.from_numpy(synth.gauss_ball(B, N, 5)).to(dev)
    feat0 = torch.randn(B, N, 16, device=dev)
    _, new_xyz = ops.furthest_point_sample(x, m)
    idx, cnt = ops.ball_query(new_xyz, x, 0.3, ns, return_cnt=True)
    goff = ops.group_offsets(cnt)
    mlp = PointwiseMLP([19, 64, 96, cl]).to(dev).train()
    rows = torch.randn(B, m, ns, 19, device=dev)
    wide = torch.randn(B, m, cl + 64 + 320, device=dev)

    def go(strided):
        mm = copy.deepcopy(mlp)
        f = feat0.clone().requires_grad_(True)
        r = rows.clone().requires_grad_(True)
        out = mm.forward_grouped(x, new_xyz, f, idx, cnt, goff, True) if grouped else mm(r, group_max=ns)
        assert out.shape == (B, m, cl)
        g = wide[..., 64:64 + cl]
        out.backward(g if strided else g.contiguous())
        return ((f if grouped else r).grad.detach(), {n: p.grad.detach() for n, p in mm.named_parameters()})

    def both():
        n0 = mlp_hip.COUNTERS["strided_gout"]
        a = go(True)
        assert mlp_hip.COUNTERS["strided_gout"] == n0 + 1
        b = go(False)
        assert mlp_hip.COUNTERS["strided_gout"] == n0 + 1
        return a, b

    a, b = _with_stack(True, both)
    # (the scatter to the points is a gather over row lists: deterministic, so everything is bit-identical)
    assert torch.equal(a[0], b[0])
    for n in a[1]:
        assert torch.equal(a[1][n], b[1][n]), n


@pytest.mark.parametrize("grouped", [False, True, "deep"])
def test_stack_with_a_96_wide_hidden_layer_runs_zero_padded(dev, grouped):
    """csrc/stack.hip runs a hidden width of 96 (the MSG part-seg encoder's [3, 64, 96, 128]) as 128 with zero weights / gamma / beta in
    the pad once the stack has >= 32768 rows: the 96 real channels see the same products, so outputs, input gradient, every parameter
    gradient (shapes of the 96-wide module) and the running statistics must equal the per-kernel path (which does not pad) to fp32
    summation order -- the two paths tile the rows differently."""
    from pointcloudlib_amd import synth
    from pointcloudlib_amd.misc import ops
    torch.manual_seed(11)
    deep = grouped == "deep"         # three 96-wide hidden layers with bias: 19 items in the forward's pad table (it held 12)
    grouped = grouped is True
    if grouped:
        B, N, m, ns = 8, 1024, 512, 32                 # 131072 grouped rows
        x = torch.from_numpy(synth.gauss_ball(B, N, 33)).to(dev)
        feat0 = torch.randn(B, N, 3, device=dev)
        _, new_xyz = ops.furthest_point_sample(x, m)
        idx, cnt = ops.ball_query(new_xyz, x, 0.3, ns, return_cnt=True)
        goff = ops.group_offsets(cnt)
        mlp = PointwiseMLP([6, 64, 96, 128]).to(dev).train()
        gout = torch.randn(B, m, 128, device=dev)
    else:
        mlp = PointwiseMLP([32, 96, 96, 96, 128] if deep else [32, 64, 96, 128], bias=True).to(dev).train()
        xin = torch.randn(40000, 32, device=dev)
        gout = torch.randn(40000, 128, device=dev)
    with torch.no_grad():
        for g in mlp.gammas:
            g.uniform_(0.5, 1.5); g[::4] *= -1.0

    def go():
        mm = copy.deepcopy(mlp)
        if grouped:
            out = mm.forward_grouped(x, new_xyz, feat0, idx, cnt, goff, True)
            xg = None
        else:
            xi = xin.clone().requires_grad_(True)
            out = mm(xi)
        out.backward(gout)
        if not grouped:
            xg = xi.grad.detach()
        return out.detach(), xg, {n: p.grad.detach() for n, p in mm.named_parameters()}, {n: b.detach().clone() for n, b in mm.named_buffers()}

    a, b = _with_stack(False, go), _with_stack(True, go)
    close = lambda u, v, tol: (u - v).abs().max().item() <= tol * max(1e-6, u.abs().max().item())
    assert close(a[0], b[0], 2e-6), (a[0] - b[0]).abs().max().item()
    if a[1] is not None:
        assert close(a[1], b[1], 1e-4)
    for n in a[2]:
        assert a[2][n].shape == b[2][n].shape, n
        assert close(a[2][n], b[2][n], 2e-4), (n, (a[2][n] - b[2][n]).abs().max().item(), a[2][n].abs().max().item())
    for n in a[3]:
        assert a[3][n].shape == b[3][n].shape and close(a[3][n].float(), b[3][n].float(), 1e-5), n
    # the per-kernel path pads the same way (mlp_hip._profiling_pad: it is what the profiling passes time): same kernels, same operands
    assert torch.equal(a[0], b[0])
    for n in a[2]:
        assert torch.equal(a[2][n], b[2][n]), n


# ---- the whole FC head as one call per direction (pcl_fc_head_*_f32) ---------------------------------------------------
def _head_mods(dev, spec, bn, bias, slope, p):
    from torch import nn
    mods = []
    for i in range(len(spec) - 1):
        last = i == len(spec) - 2
        mods.append(nn.Linear(spec[i], spec[i + 1], bias=bias or last))
        if not last:
            if bn:
                mods.append(nn.BatchNorm1d(spec[i + 1]))
            mods.append(nn.ReLU() if slope == 0.0 else nn.LeakyReLU(slope))
            if p is not None:
                mods.append(nn.Dropout(p))
    return nn.Sequential(*mods).to(dev)


@pytest.mark.parametrize("spec,R,bn,bias,slope", [([1024, 512, 256, 40], 32, True, False, 0.0), ([2048, 512, 256, 40], 16, True, True, 0.2),
                                                  ([64, 40], 5, False, True, 0.0), ([100, 30, 7], 64, True, True, 0.2),
                                                  ([48, 96, 96, 24, 10], 33, True, False, 0.0)])
@pytest.mark.parametrize("training", [True, False])
def test_head_stack_equals_per_layer_path(dev, spec, R, bn, bias, slope, training):
    """pcl_fc_head_*_f32 chains the kernels of pcl_head_layer_*_f32: with dropout off, the output, the running statistics and
    the last layer's parameter gradients are BIT-identical to the per-layer path; what passes through a dX kernel (fp32
    atomics over the column splits) agrees to summation order."""
    from pointcloudlib_amd.misc import head
    torch.manual_seed(5)
    seq = _head_mods(dev, spec, bn, bias, slope, 0.0)
    seq.train(training)
    x0 = torch.randn(R, spec[0], device=dev)
    gout = torch.randn(R, spec[-1], device=dev)
    res = []
    for flag in (False, True):
        s2 = copy.deepcopy(seq)
        old, head.USE_STACK = head.USE_STACK, flag
        try:
            x = x0.clone().requires_grad_(True)
            out = head.fc_head(s2, x)
            out.backward(gout)
        finally:
            head.USE_STACK = old
        res.append((out.detach(), x.grad.detach(), [p.grad.detach() for p in s2.parameters()], [b.detach().clone() for b in s2.buffers()]))
    a, b = res
    assert torch.equal(a[0], b[0])
    gs = max(u.abs().max().item() for u in a[2])               # (a conv bias under BatchNorm has an exactly-zero gradient: judge by the model's scale)
    assert (a[1] - b[1]).abs().max().item() <= 1e-5 * max(1e-6, a[1].abs().max().item())
    for u, v in zip(a[2], b[2]):
        assert (u - v).abs().max().item() <= 1e-5 * u.abs().max().item() + 2e-7 * gs
    n_last = 2                                                 # the last Linear always carries a bias here: (W, b) are its parameters
    for u, v in zip(a[2][-n_last:], b[2][-n_last:]):
        assert torch.equal(u, v)
    for u, v in zip(a[3], b[3]):
        assert torch.equal(u.float(), v.float())


@pytest.mark.parametrize("p", [0.5, 0.2])
def test_head_stack_dropout(dev, p):
    """Dropout inside the head kernels: inverted dropout on a layer's output with the keep decision recomputed in backward.
    A linear probe layer (identity weight) behind the dropped layer makes the mask visible."""
    from torch import nn
    from pointcloudlib_amd.misc import head
    torch.manual_seed(9)
    R, K, N = 32, 96, 256
    lin1, lin2 = nn.Linear(K, N).to(dev), nn.Linear(N, N, bias=False).to(dev)
    with torch.no_grad():
        lin2.weight.copy_(torch.eye(N))
    mods = [lin1, nn.Dropout(p), lin2]
    for m in mods:
        m.train()
    x = torch.randn(R, K, device=dev, requires_grad=True)
    out = head.fc_head(mods, x)
    y = torch.nn.functional.linear(x.detach(), lin1.weight, lin1.bias)
    mask = out.detach() != 0
    keep = mask.float().mean().item()
    assert abs(keep - (1 - p)) < 0.03, keep
    assert torch.allclose(out.detach()[mask], (y / (1 - p))[mask], rtol=1e-5, atol=1e-6)
    gout = torch.randn(R, N, device=dev)
    out.backward(gout)
    want_dx = (gout * mask / (1 - p)) @ lin1.weight.detach()
    assert torch.allclose(x.grad, want_dx, rtol=1e-4, atol=1e-5)
    want_dw = (gout * mask / (1 - p)).t() @ x.detach()
    assert torch.allclose(lin1.weight.grad, want_dw, rtol=1e-4, atol=1e-5)
    out2 = head.fc_head(mods, x.detach())                # another call draws another mask
    assert ((out2 != 0) != mask).float().mean().item() > 0.1
    for m in mods:
        m.eval()
    out3 = head.fc_head(mods, x.detach())                # evaluation mode: identity
    assert torch.allclose(out3, y, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("P,C,offset", [(4096, 48, 0.0), (100000, 96, 3.0), (33, 7, 0.0), (20000, 288, -2.0), (5, 512, 0.5)])
def test_bn_rows_matches_fp64(dev, P, C, offset):
    """The stand-alone BatchNorm over rows (PointCNN's BatchNorm after an activation; csrc/mlp.hip bn_rows_*) against fp64
    PyTorch: output 1e-5 of its scale, input / gamma / beta gradients 1e-5 relative to their max-norm (+ the model scale for
    analytically small ones), running statistics by Jittor's rule (biased variance, momentum as given)."""
    from pointcloudlib_amd.misc.layers import batch_norm_train
    torch.manual_seed(P + C)
    x0 = (torch.randn(P, C, dtype=torch.float64) * torch.rand(C, dtype=torch.float64).add(0.2) + offset).relu()
    g0, b0 = torch.rand(C, dtype=torch.float64) + 0.5, torch.randn(C, dtype=torch.float64)
    gout = torch.randn(P, C, dtype=torch.float64)
    x64, g64, b64 = x0.clone().requires_grad_(True), g0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    want = torch.nn.functional.batch_norm(x64, None, None, g64, b64, True, 0.0, 1e-5)
    want.backward(gout)
    x, g, b = (t.float().to(dev).requires_grad_(True) for t in (x0, g0, b0))
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    got = batch_norm_train(x, g, b, rm, rv, True, momentum=0.9, eps=1e-5)
    got.backward(gout.float().to(dev))
    assert (got.detach().cpu().double() - want.detach()).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())
    gs = max(t.grad.abs().max().item() for t in (x64, g64, b64))
    for name, a, r in (("dx", x.grad, x64.grad), ("dgamma", g.grad, g64.grad), ("dbeta", b.grad, b64.grad)):
        assert (a.cpu().double() - r).abs().max().item() <= 2e-5 * r.abs().max().item() + 1e-6 * gs, name
    mean, var = x0.mean(0), x0.var(0, unbiased=False)
    assert torch.allclose(rm.cpu().double(), 0.9 * mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(rv.cpu().double(), 1.0 + 0.9 * (var - 1.0), rtol=1e-5, atol=1e-6)


@pytest.fixture(params=[0, 1], ids=["fp32-mfma", "bf16-planes"])
def matrix_form(request):
    from pointcloudlib_amd import _lib
    _lib.lib().pcl_set_matrix_form(request.param)
    yield request.param
    _lib.lib().pcl_set_matrix_form(0)


@pytest.mark.parametrize("spec,rows,bias,slope", [([8, 64, 64], 40001, False, 0.0), ([8, 64, 128], 33000, True, 0.2), ([12, 128, 128], 70007, False, 0.0),
                                                  ([12, 128, 256], 36864, False, 0.0), ([8, 64, 64, 128], 50000, False, 0.0),
                                                  ([64, 128], 33333, False, 0.2), ([128, 256, 64], 40000, True, 0.0)])      # first layer = plain input
def test_resident_weight_forward_kernel_against_fp64(dev, matrix_form, spec, rows, bias, slope):
    """linear_fwd_res_kernel (hidden layers of the set-abstraction shapes at >= 32768 rows: weight slab resident in LDS, 8 waves,
    double-buffered row image): module output and running statistics against fp64 PyTorch, ragged last tile, bias, LeakyReLU,
    Cout = 256 as two slabs."""
    torch.manual_seed(rows)
    mlp = PointwiseMLP(spec, bias=bias, slope=slope).to(dev).train()
    with torch.no_grad():
        for g in mlp.gammas:
            g.uniform_(0.5, 1.5); g[::3] *= -1.0
    x = torch.randn(rows, spec[0], device=dev)
    ref = copy.deepcopy(mlp).cpu().double()
    ref.backend = "torch"
    want = ref(x.cpu().double())
    got = mlp(x)
    assert (got.detach().cpu().double() - want.detach()).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())
    for (n, b), (_, r) in zip(mlp.named_buffers(), ref.named_buffers()):
        assert torch.allclose(b.cpu().double(), r, rtol=1e-5, atol=1e-6), n


# ---- narrow stacks (csrc/narrow.hip): PointConv's WeightNet 3-8-8-16 and DensityNet 1-8-8-1, recomputed per BatchNorm pass ----
def _run_nograd(module, x, gout, backend):
    module.backend = backend
    module.zero_grad()
    out = module(x)
    out.backward(gout)
    return (out.detach(), {n: p.grad.detach().clone() for n, p in module.named_parameters()},
            {n: b.detach().clone() for n, b in module.named_buffers()})


def _narrow_plan_bytes(m, rows):
    from pointcloudlib_amd.misc import mlp_hip
    return mlp_hip._stack_plan(m, rows, m.spec[0], 0, False, None, False, 0).save_bytes


NARROW_CASES = [(spec, bias, slope, off, spr, rows)
                for spec, bias, slope, off, spr in [([3, 8, 8, 16], True, 0.0, 0.0, 1.0), ([1, 8, 8, 1], True, 0.0, 0.0, 1.0),
                                                    ([3, 8, 8, 16], False, 0.2, 0.0, 1.0), ([1, 8, 8, 1], True, 0.0, 5.0, 0.01)]
                for rows in (2, 300, 5000, 70001)]
# |mean| / std = 300 on three input channels: every fp32 path (PyTorch's included) rounds the first layer's output at 6e-5 of its
# spread, so at a few thousand rows some ReLU mask differs from the fp64 run and moves the batch sums by 1e-3 .. 1e-2 of their
# size, for the GEMM path and PyTorch alike (tools/dbg/narrow_err.py) -- row counts where no mask flips only
NARROW_CASES += [([3, 8, 8, 16], True, 0.0, 0.3, 1e-3, rows) for rows in (2, 300, 1000)]


@pytest.mark.parametrize("spec,bias,slope,offset,spread,rows", NARROW_CASES)
def test_narrow_stack_matches_fp64(dev, spec, bias, slope, offset, spread, rows):
    """The recompute-per-pass kernels against the PyTorch composition in fp64: output, every parameter gradient, running
    statistics; also with |mean| >> std inputs (DensityNet reads near-constant densities): never worse than 4x PyTorch's own fp32
    path against the same truth.  The descriptor must have taken the narrow path (768 bytes live from forward to backward)."""
    torch.manual_seed(7 + rows + spec[0])
    m64 = PointwiseMLP(spec, bias=bias, slope=slope).double()
    with torch.no_grad():
        for g, b in zip(m64.gammas, m64.betas):
            g.uniform_(0.5, 1.5); b.uniform_(-0.3, 0.3)
        m64.gammas[1][::3] *= -1.0
    x64 = offset + spread * torch.randn(rows, spec[0], dtype=torch.float64)
    g64 = torch.randn(rows, spec[-1], dtype=torch.float64)
    ref = _run_nograd(copy.deepcopy(m64), x64, g64, "torch")
    m32 = copy.deepcopy(m64).float().to(dev).train()
    import os
    if os.environ.get("PCL_NARROW") != "0":
        assert _narrow_plan_bytes(m32, rows) == 3 * 4 * 16 * 4
    x32, g32 = x64.float().to(dev), g64.float().to(dev)
    t32 = _run_nograd(copy.deepcopy(m32), x32, g32, "torch")
    h32 = _run_nograd(copy.deepcopy(m32), x32, g32, "hip")

    def err(a, b):
        return (a.double().cpu() - b).abs().max().item()

    scale = max(1.0, ref[0].abs().max().item())
    e_h, e_t = err(h32[0], ref[0]), err(t32[0], ref[0])
    assert e_h <= max(1e-5 * scale, 4 * e_t), f"features: hip err {e_h:.3e} (torch fp32 {e_t:.3e}), scale {scale:.3f}"
    for name in ref[1]:
        if "biases" in name:
            assert h32[1][name].abs().max().item() == 0.0
            continue
        gs = max(1e-6, ref[1][name].abs().max().item())
        e_h, e_t = err(h32[1][name], ref[1][name]), err(t32[1][name], ref[1][name])
        assert e_h <= max(1e-4 * gs, 4 * e_t), f"{name}: hip {e_h:.3e} torch {e_t:.3e} scale {gs:.3e}"
    for name in ref[2]:
        assert err(h32[2][name], ref[2][name]) <= max(1e-5 * max(1.0, ref[2][name].abs().max().item()), 4 * err(t32[2][name], ref[2][name])), name


@pytest.mark.parametrize("spec,rows", [([3, 8, 8, 16], 32 * 512 * 32), ([1, 8, 8, 1], 32 * 1024), ([3, 8, 8, 16], 32 * 128 * 64 + 5)])
def test_narrow_stack_equals_the_gemm_path_at_size(dev, spec, rows):
    """PointConv's row counts: the narrow kernels against the library's GEMM / BatchNorm kernels (per-kernel path) on the same
    inputs -- the same arithmetic per element up to the summation order of the dot products and of the batch sums."""
    from pointcloudlib_amd.misc import mlp_hip
    torch.manual_seed(5)
    m = PointwiseMLP(spec, bias=True).to(dev).train()
    x = torch.randn(rows, spec[0], device=dev) * 0.3
    g = torch.randn(rows, spec[-1], device=dev)
    a = _run_nograd(copy.deepcopy(m), x, g, "hip")
    with mlp_hip.per_kernel_path():
        b = _run_nograd(copy.deepcopy(m), x, g, "hip")
    scale = max(1.0, b[0].abs().max().item())
    assert (a[0] - b[0]).abs().max().item() <= 1e-5 * scale
    flips = ((a[0] > 0) != (b[0] > 0)).float().mean().item()
    assert flips < 1e-5
    # Gradients: the two fp32 paths round a pre-activation differently by ~1e-7, so about one ReLU mask per 10^7 elements differs
    # between them (8M masked elements here): one flipped row moves a batch sum by that row's term, amplified by the layers below
    # it -- 3e-3 of the gradient's max-norm covers a handful (against fp64 the narrow kernels are the closer of the two:
    # tools/dbg/narrow_err.py).  DensityNet's first-layer weight gradient is exactly ZERO in exact arithmetic (one input channel:
    # BatchNorm removes scale and shift of the affine map), both paths return rounding noise of sums whose terms are O(1).
    for n in b[1]:
        gs = max(1e-6, b[1][n].abs().max().item())
        d = (a[1][n] - b[1][n]).abs().max().item()
        if spec[0] == 1 and n == "weights.0":
            assert max(a[1][n].abs().max().item(), b[1][n].abs().max().item()) <= 1e-6 * rows, (n, a[1][n], b[1][n])
            continue
        assert d <= 3e-3 * gs + 1e-7, (n, d, gs)
    for n in b[2]:
        assert torch.allclose(a[2][n], b[2][n], rtol=1e-5, atol=1e-6), n


# ---- fp32 operands as three bf16 planes on the bf16 matrix pipe (round 4) ----
def _raw_forward(dev, x, w, sc, sh, slope, split):
    from pointcloudlib_amd import _lib
    from pointcloudlib_amd.misc.mlp_hip import _P, _stream
    P, cin = x.shape
    cout = w.shape[0]
    rows = _lib.size_query("pcl_mlp_stat_rows", P, cout, 0)
    y = torch.empty(P, cout, device=dev)
    stats = torch.empty(rows, 2, cout, device=dev, dtype=torch.float64)
    _lib.lib().pcl_set_matrix_form(int(split))
    try:
        _lib.call("pcl_linear_fwd_rows_f32", _P(x), _P(w), None, _P(sc), _P(sh), slope, P, cin, cout, _P(y), _P(stats), None, None, _stream())
    finally:
        _lib.lib().pcl_set_matrix_form(0)
    torch.cuda.synchronize()
    return y, stats.sum(0)


@pytest.mark.parametrize("cin,cout,rows,act", [(64, 64, 40001, True), (64, 128, 65536, True), (128, 128, 33333, False), (128, 256, 50000, True)])
@pytest.mark.parametrize("spread", [1.0, 1e4])
def test_split_gemm_error_vs_fp64(dev, cin, cout, rows, act, spread):
    """The nine-product bf16 form of an fp32 GEMM against the fp32 MFMA form, both against fp64: every partial product of the
    split form is exact, so its error is the accumulator's rounding alone -- the same size as the fp32 MFMA form's (measured: 0.7-1.08 x
    of its mean error), on operands spanning eight orders of magnitude as well (`spread`: per-row scales 1 .. 1e4 and
    per-column weights 1 .. 1e-4, where a dropped low-order plane would show)."""
    torch.manual_seed(cin * 7 + cout)
    x = torch.randn(rows, cin, device=dev) * torch.logspace(0, float(np.log10(spread)), rows, device=dev)[torch.randperm(rows, device=dev)].unsqueeze(1)
    w = torch.randn(cout, cin, device=dev) / cin ** 0.5 * torch.logspace(0, -float(np.log10(spread)), cin, device=dev).unsqueeze(0)
    sc = (torch.rand(cin, device=dev) + 0.5) if act else None
    sh = (torch.randn(cin, device=dev) * 0.1) if act else None
    z = x.double()
    if act:
        z = torch.relu(torch.addcmul(sh, sc, x)).double()      # the kernel's own fp32 fma + max: the operand both forms multiply
    want = z @ w.double().t()
    mag = z.abs() @ w.double().abs().t()                          # sum_k |a_k b_k|: what rounding errors scale with
    errs = {}
    for split in (1, 0):
        y, st = _raw_forward(dev, x, w, sc, sh, 0.0, split)
        e = ((y.double() - want).abs() / mag.clamp_min(1e-300))
        errs[split] = (e.max().item(), e.mean().item())
        assert torch.allclose(st[0], y.double().sum(0), rtol=1e-6, atol=1e-4 * spread)
    assert errs[1][0] <= 1.5e-6, errs                             # a few fp32 ulps of the magnitude sum, K <= 128 (measured 7-8e-7)
    assert errs[1][1] <= 1.25 * errs[0][1], errs                  # mean error: that of the fp32 MFMA form (measured 0.7-1.08 x)


@pytest.fixture
def split_gemms():
    """Every staged GEMM with K >= 16 on the bf16-plane form (pcl_set_matrix_form bit 1), the fused backward off so that dX / dW
    go through the staged kernels too."""
    from pointcloudlib_amd import _lib
    from pointcloudlib_amd.misc import mlp_hip
    old = mlp_hip._FUSED_BWD
    _lib.lib().pcl_set_matrix_form(2 | (16 << 8))
    mlp_hip._FUSED_BWD = False
    yield
    mlp_hip._FUSED_BWD = old
    _lib.lib().pcl_set_matrix_form(0)


SPLIT_CASES = [c for c in CASES if max(c[0][:-1]) >= 16] + FB_CASES + [
    ([512, 1024], (2, 2048), None, False, 0.0),             # DGCNN's conv5 shape at 4096 rows (128x128 tiles, 16 k steps)
    ([64, 128, 1024], (4, 512), None, False, 0.0),          # PointNet's trunk
    ([256, 512, 1024], (3, 1, 128), 128, False, 0.0),       # the GroupAll level without the xyz columns (vector path, 64-row tiles)
]


@pytest.mark.parametrize("spec,lead,ns,bias,slope", SPLIT_CASES)
def test_split_staged_gemms_match_reference(dev, split_gemms, spec, lead, ns, bias, slope):
    """linear_nt_kernel with bf16-plane operands (forward, dX with the dy transform and the masked epilogue, ragged / max-pooled /
    sparse modes): the same fp64 checks as the fp32 MFMA form."""
    _check_against_fp64(dev, spec, lead, ns, bias, slope)


# ---- kernel-selection lab switches (pcl_set_kernel_paths): every alternative path computes the same layer ----
@pytest.mark.parametrize("off", ["fwd_resident", "fused_backward", "narrow_stacks"])
def test_kernel_path_switches_select_equivalent_kernels(dev, off):
    """Each switch routes a stack to the other kernel family (staged forward instead of the resident-weight one, separate dX / dW
    instead of the fused backward, GEMM kernels instead of the recompute-per-pass narrow stack): same outputs and gradients up to
    fp32 summation order.  The switches are C calls -- the library itself reads no environment variable."""
    from pointcloudlib_amd import _lib
    torch.manual_seed(11)
    if off == "narrow_stacks":
        spec, rows = [3, 8, 8, 16], 40000
    else:
        spec, rows = [16, 64, 128, 256], 36000
    m = PointwiseMLP(spec, bias=(off == "narrow_stacks")).to(dev).train()
    x = torch.randn(rows, spec[0], device=dev) * 0.5
    g = torch.randn(rows, spec[-1], device=dev)
    a = _run_nograd(copy.deepcopy(m), x, g, "hip")
    args = {"fwd_resident": (0, -1, -1), "narrow_stacks": (-1, 0, -1), "fused_backward": (-1, -1, 0)}[off]
    _lib.lib().pcl_set_kernel_paths(*args)
    try:
        b = _run_nograd(copy.deepcopy(m), x, g, "hip")
    finally:
        _lib.lib().pcl_set_kernel_paths(1, 1, 1)
    scale = max(1.0, a[0].abs().max().item())
    assert (a[0] - b[0]).abs().max().item() <= 1e-5 * scale
    for n in a[1]:
        gs = max(1e-6, a[1][n].abs().max().item())
        # a handful of ReLU masks differ between two fp32 paths (see test_narrow_stack_equals_the_gemm_path_at_size); at 40 000 rows of
        # 8-wide layers one flipped row weighs 7e-3 of a weight gradient's max-norm
        tol = 2e-2 if off == "narrow_stacks" else 3e-3
        assert (a[1][n] - b[1][n]).abs().max().item() <= tol * gs + 1e-7, n
    for n in a[2]:
        assert torch.allclose(a[2][n], b[2][n], rtol=1e-5, atol=1e-6), n


# ---- weight gradients on the library's side stream (pcl_set_stack_overlap): same kernels, any interleaving ----
@pytest.mark.parametrize("spec,lead,ns,bias", [([259, 256, 512, 1024], (32, 1, 128), 128, False),     # the GroupAll level at full size
                                               ([1664, 256, 256], (16, 128), None, True),               # part-seg decoder fp3 (no max, bias)
                                               ([20, 96, 32], (2, 300), None, True)])
def test_side_stream_weight_gradients_equal_the_serial_order(dev, spec, lead, ns, bias):
    """pcl_mlp_stack_bwd_f32 forks the dW launches of few-row plain stacks to a second stream and joins at the end of the call
    (csrc/stack.hip: side_dw): outputs and every gradient must be BIT-identical to the serial order, call after call."""
    from pointcloudlib_amd import _lib
    from pointcloudlib_amd.misc import mlp_hip
    L = _lib.lib()
    torch.manual_seed(11)
    mlp = PointwiseMLP(spec, bias=bias, slope=0.0).to(dev).train()
    x = torch.randn(*lead, spec[0], device=dev)
    gshape = (lead[:-1] if ns else lead) + (spec[-1],)
    gout = torch.randn(*gshape, device=dev)
    prev = L.pcl_get_stack_overlap()
    try:
        L.pcl_set_stack_overlap(0, -1); mlp_hip._PLANS.clear()          # (buffer sizes depend on the mode)
        ref = run(copy.deepcopy(mlp), x, ns, gout, "auto")
        L.pcl_set_stack_overlap(1, -1); mlp_hip._PLANS.clear()
        for _ in range(4):
            got = run(copy.deepcopy(mlp), x, ns, gout, "auto")
            # consume the gradients right away on the caller's stream: the join must order them behind the side stream's writes
            assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])
            for n in ref[2]:
                assert torch.equal(ref[2][n], got[2][n]), n
    finally:
        L.pcl_set_stack_overlap(prev, -1); mlp_hip._PLANS.clear()


@pytest.mark.parametrize("spec,lead,ns,bias", [([259, 256, 512, 1024], (32, 1, 128), 128, False),     # the GroupAll level at full size (sparse max gradient)
                                               ([1664, 256, 256], (16, 128), None, True),               # part-seg decoder fp3 (dense, bias)
                                               ([131, 128, 96, 64], (3, 50, 8), 8, False)])             # 96 = 3 x 32 channels, ragged row slabs
def test_few_row_backward_matches_the_staged_kernels(dev, spec, lead, ns, bias):
    """pcl_set_fewrow_backward(1): BatchNorm-backward constants + dy in one launch (pcl_bn_bwd_dy_f32), dX on the fragment kernel and dW on
    the staged kernel reading the formed dy.  dgamma / dbeta come from the same sums in the same order (bit-identical); the GEMMs sum in
    another order: 1e-4 of the gradient's max-norm (this file's gradient tolerance).  The per-stack and the per-kernel host path must launch the same kernels (bit-identical)."""
    from pointcloudlib_amd import _lib
    from pointcloudlib_amd.misc import mlp_hip
    L = _lib.lib()
    torch.manual_seed(5)
    mlp = PointwiseMLP(spec, bias=bias, slope=0.0).to(dev).train()
    x = torch.randn(*lead, spec[0], device=dev)
    gshape = (lead[:-1] if ns else lead) + (spec[-1],)
    gout = torch.randn(*gshape, device=dev)
    prev = L.pcl_get_fewrow_backward()
    def fresh():
        _lib.size_query.cache_clear(); mlp_hip._PLANS.clear()
    try:
        L.pcl_set_fewrow_backward(0); fresh()
        ref = run(copy.deepcopy(mlp), x, ns, gout, "auto")
        L.pcl_set_fewrow_backward(1); fresh()
        assert L.pcl_mlp_fewrow_layer(x.numel() // spec[0], spec[-1], spec[-2], 0) == 1
        got = _with_stack(True, lambda: run(copy.deepcopy(mlp), x, ns, gout, "auto"))
        per = _with_stack(False, lambda: run(copy.deepcopy(mlp), x, ns, gout, "auto"))
    finally:
        L.pcl_set_fewrow_backward(prev); fresh()
    assert torch.equal(ref[0], got[0])
    L_ = len(spec) - 1
    for n in ref[2]:
        assert torch.equal(got[2][n], per[2][n]), n
        if n.startswith(("gammas", "betas")) and n.endswith(str(L_ - 1)):
            assert torch.equal(ref[2][n], got[2][n]), n          # the last layer's sums come from the same upstream gradient
        err = (ref[2][n] - got[2][n]).abs().max().item()
        assert err <= 1e-4 * max(1e-30, ref[2][n].abs().max().item()) + 1e-7, (n, err)
    assert torch.equal(got[1], per[1])
    assert (ref[1] - got[1]).abs().max().item() <= 1e-4 * ref[1].abs().max().item() + 1e-7


@pytest.mark.parametrize("B,N,m,ns,C1", [(4, 512, 96, 32, 128), (3, 200, 40, 16, 64), (2, 1024, 128, 24, 256), (2, 2048, 64, 32, 128)])
def test_scatter_as_a_gather_over_the_points_row_lists(dev, B, N, m, ns, C1):
    """pcl_group_rows_transpose_i32 (every source point's rows, ascending) against a torch sort, and pcl_group_linear_bwd_gather_f32 (the
    folded first layer's backward walking the rows by source point: dUf written once, no atomics) against an fp64 index_add and against
    pcl_group_linear_bwd_f32 (fp32 atomics): same sums to fp32 rounding, run-to-run identical, unreferenced points exactly zero."""
    import ctypes
    from pointcloudlib_amd import _lib
    L = _lib.lib()
    _p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(B * N + C1)
    assert L.pcl_group_rows_transpose_supported(N, m, ns) and L.pcl_group_linear_bwd_gather_supported(C1)
    cnt = torch.randint(1, ns + 1, (B * m,), device=dev)
    goff = torch.zeros(B * m + 1, dtype=torch.int32, device=dev); goff[1:] = torch.cumsum(cnt, 0).int()
    R = int(goff[-1].item())
    grp = torch.repeat_interleave(torch.arange(B * m, device=dev), cnt)
    # distinct sources within a group (what ball query / k-NN produce), from the first half of the cloud: the other half is never referenced
    pick = torch.rand(B * m, max(1, N // 2), device=dev).argsort(1)[:, :ns]                    # a random subset per group
    slot = torch.arange(R, device=dev) - goff[:-1].long()[grp]
    src = ((grp // m) * N + pick[grp, slot]).int()
    loc = torch.randn(R, 4, device=dev); loc[:, 3] = torch.randint(1, 4, (R,), device=dev).float()
    dU, Y = torch.randn(R, C1, device=dev), torch.randn(R, C1, device=dev)
    a, k1, k2, mu = (torch.randn(C1, device=dev) for _ in range(4))
    rows_blk = L.pcl_group_linear_stat_rows(B, m)
    in_off = torch.empty(B * N + 1, dtype=torch.int32, device=dev)
    in_rows = torch.empty(R, dtype=torch.int32, device=dev)
    _lib.call("pcl_group_rows_transpose_i32", _p(src), _p(goff), B, N, m, ns, _p(in_off), _p(in_rows), st)
    order = torch.sort(src.long() * (R + 1) + torch.arange(R, device=dev)).indices.int()
    ref_off = torch.zeros(B * N + 1, dtype=torch.int64, device=dev)
    ref_off[1:] = torch.cumsum(torch.bincount(src.long(), minlength=B * N), 0)
    assert torch.equal(in_off.long(), ref_off) and torch.equal(in_rows, order)
    dy = a * dU - loc[:, 3:4] * (k1 + k2 * (Y - mu))
    ref = torch.zeros(B * N, C1, device=dev, dtype=torch.float64).index_add_(0, src.long(), dy.double())
    refw = torch.einsum("rc,rd->cd", dy.double(), loc[:, :3].double())
    outs = []
    for which in ("atomics", "gather", "gather"):
        dUf = torch.full((B * N, C1), float("nan"), device=dev)
        dWx = torch.empty(rows_blk, C1, 3, device=dev)
        dW0 = torch.empty(C1, 3, device=dev)
        if which == "atomics":
            _lib.call("pcl_group_linear_bwd_f32", _p(loc), None, 0, _p(dU), _p(Y), _p(a), _p(k1), _p(k2), _p(mu), _p(src), _p(goff[B * m:]), B, N, C1,
                      _p(dUf), _p(dWx), None, _p(dW0), 3, 0, st)
        else:
            _lib.call("pcl_group_linear_bwd_gather_f32", _p(loc), _p(dU), _p(Y), _p(a), _p(k1), _p(k2), _p(mu), _p(in_off), _p(in_rows), B, N, C1,
                      _p(dUf), _p(dWx), _p(dW0), 3, st)
        outs.append((dUf, dW0))
    sc, scw = ref.abs().max().item(), refw.abs().max().item()
    for dUf, dW0 in outs:
        assert (dUf.double() - ref).abs().max().item() <= 1e-5 * sc
        assert (dW0.double() - refw).abs().max().item() <= 1e-4 * scw
    assert torch.equal(outs[1][0], outs[2][0]) and torch.equal(outs[1][1], outs[2][1])
    assert (outs[1][0][ref == 0] == 0).all()


@pytest.mark.parametrize("spec,lead,ns,bias", [([259, 256, 512, 1024], (32, 1, 128), 128, False),     # the GroupAll level at full size (sparse max gradient, K = 259 scalar path)
                                               ([1664, 256, 256], (16, 128), None, True),               # part-seg decoder fp3 (dense gradient, bias)
                                               ([128, 128, 128, 128], (3, 700), None, False),           # 2 100 rows, every layer eligible
                                               ([20, 96, 32], (2, 300), None, True)])                   # nothing eligible (widths <= 64 / padded): the old launches
def test_backward_pair_launch_equals_the_separate_launches(dev, spec, lead, ns, bias):
    """Round 6: both backward GEMMs of a few-row layer run in ONE launch and the split-K tile sum rides in the launch that forms the layer
    below's constants (csrc/mlp.hip: linear_bwd_pair_kernel, pcl_linear_bwd_pair_finish_f32).  Same kernel bodies, same reduction order:
    outputs and every gradient must be BIT-identical to the four-launch form (pcl_set_bwd_pair(0))."""
    from pointcloudlib_amd import _lib
    from pointcloudlib_amd.misc import mlp_hip
    L = _lib.lib()
    torch.manual_seed(12)
    mlp = PointwiseMLP(spec, bias=bias, slope=0.0).to(dev).train()
    x = torch.randn(*lead, spec[0], device=dev)
    gshape = (lead[:-1] if ns else lead) + (spec[-1],)
    gout = torch.randn(*gshape, device=dev)
    prev = L.pcl_get_bwd_pair()
    try:
        L.pcl_set_bwd_pair(0); mlp_hip._PLANS.clear()
        ref = run(copy.deepcopy(mlp), x, ns, gout, "auto")
        L.pcl_set_bwd_pair(1); mlp_hip._PLANS.clear()
        got = run(copy.deepcopy(mlp), x, ns, gout, "auto")
        assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])
        for n in ref[2]:
            assert torch.equal(ref[2][n], got[2][n]), n
    finally:
        L.pcl_set_bwd_pair(prev); mlp_hip._PLANS.clear()


@pytest.mark.parametrize("spec,lead,ns", [([16, 64, 128, 64], (3, 44444), None),            # dense 128 x 64 at size (2-9 tiles per workgroup)
                                          ([5, 64, 64, 64, 128], (7, 900, 16), 16),         # sparse 128 x 64 under a max pool, 100 800 rows
                                          ([4, 64, 128, 64], (1, 65), None)])               # 65 rows: one 64-row tile + 1 row
def test_two_image_fused_backward_equals_the_one_image_form(dev, spec, lead, ns):
    """Round 6: the 128 x 64 fused backward walks 64-row tiles over two LDS images with the waves in two roles (pcl_set_fb_two_images).
    Same arithmetic per element; dW and the BatchNorm sums differ from the one-image form in summation order only."""
    from pointcloudlib_amd import _lib
    from pointcloudlib_amd.misc import mlp_hip
    L = _lib.lib()
    torch.manual_seed(13)
    mlp = PointwiseMLP(spec, bias=False, slope=0.0).to(dev).train()
    x = torch.randn(*lead, spec[0], device=dev)
    gshape = (lead[:-1] if ns else lead) + (spec[-1],)
    gout = torch.randn(*gshape, device=dev)
    prev = L.pcl_get_fb_two_images()
    try:
        L.pcl_set_fb_two_images(0); mlp_hip._PLANS.clear()
        ref = run(copy.deepcopy(mlp), x, ns, gout, "auto")
        L.pcl_set_fb_two_images(1); mlp_hip._PLANS.clear()
        got = run(copy.deepcopy(mlp), x, ns, gout, "auto")
        assert torch.equal(ref[0], got[0])
        for name, a, b in [("x", ref[1], got[1])] + [(n, ref[2][n], got[2][n]) for n in ref[2]]:
            scale = a.abs().max().item()
            assert (a - b).abs().max().item() <= 2e-5 * max(scale, 1e-6) + 1e-7, (name, (a - b).abs().max().item(), scale)
    finally:
        L.pcl_set_fb_two_images(prev); mlp_hip._PLANS.clear()


@pytest.mark.parametrize("P,Cout,Cin,sparse,masked,w_shift", [(4096, 512, 256, False, True, 0),      # a GroupAll-level layer, dense gradient
                                                              (4096, 1024, 512, True, True, 0),      # its last layer: sparse max gradient (ns = 128)
                                                              (700, 128, 128, False, False, 0),      # input gradient without a mask (a stack's first layer)
                                                              (2048, 256, 128, False, True, 1)])     # W one float off a 16-byte boundary: the two bodies on different paths
def test_pair_entry_point_against_the_two_entry_points(dev, P, Cout, Cin, sparse, masked, w_shift):
    """pcl_linear_bwd_pair_f32 + pcl_linear_bwd_pair_finish_f32 through the C ABI against pcl_linear_bwd_dw_rows_f32 (+ its reduce) and
    pcl_linear_bwd_dx_rows_f32: dW, dUprev and the BatchNorm-backward sums bit for bit -- also when only one of the two bodies can take
    its vector path (a weight matrix that is not 16-byte aligned: the entry point then issues the two plain launches itself)."""
    import ctypes
    from pointcloudlib_amd import _lib
    L = _lib.lib()
    _p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.pcl_linear_bwd_pair_supported(P, Cout, Cin, 0) == 1
    torch.manual_seed(21)
    ns = 128
    Y = torch.randn(P, Cout, device=dev)
    a, k1, k2, mu = (torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1, torch.randn(Cout, device=dev) * 0.1, torch.randn(Cout, device=dev) * 0.1)
    if sparse:
        G = P // ns
        arg = torch.randint(0, ns, (G, Cout), device=dev, dtype=torch.int32)
        gz = torch.randn(G, Cout, device=dev)
        dU = None
    else:
        arg = gz = None
        dU = torch.randn(P, Cout, device=dev)
    Wbuf = torch.randn(Cout * Cin + 4, device=dev)
    W = Wbuf[w_shift:w_shift + Cout * Cin].view(Cout, Cin)
    Xprev = torch.randn(P, Cin, device=dev)
    psc, psh = (torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.2) if masked else (None, None)
    srows = L.pcl_mlp_stat_rows(P, Cin, 1)
    wsb = L.pcl_linear_bwd_dw_workspace_bytes(P, Cout, Cin)

    def fresh():
        return (torch.full((P, Cin), float("nan"), device=dev), torch.zeros(srows, 2, Cin, dtype=torch.float64, device=dev),
                torch.empty(wsb // 4, device=dev), torch.full((Cout, Cin), float("nan"), device=dev))

    dXa, sta, wsa, dWa = fresh()
    rc = L.pcl_linear_bwd_dw_rows_f32(_p(dU), _p(Y), _p(a), _p(k1), _p(k2), _p(mu), _p(arg), _p(gz), ns, _p(Xprev), _p(psc), _p(psh), ctypes.c_float(0.0), P, Cout, Cin,
                                      _p(dWa), _p(wsa), wsb, None, None, 0, st)
    assert rc == 0, L.pcl_last_error()
    rc = L.pcl_linear_bwd_dx_rows_f32(_p(dU), _p(Y), _p(a), _p(k1), _p(k2), _p(mu), _p(arg), _p(gz), ns, _p(W), P, Cout, Cin, _p(Xprev) if masked else None, _p(psc), _p(psh),
                                      ctypes.c_float(0.0), _p(dXa), _p(sta) if masked else None, None, None, 0, 0, st)
    assert rc == 0, L.pcl_last_error()
    dXb, stb, wsb_t, dWb = fresh()
    rc = L.pcl_linear_bwd_pair_f32(_p(dU), _p(Y), _p(a), _p(k1), _p(k2), _p(mu), _p(arg), _p(gz), ns, _p(W), _p(Xprev), _p(psc), _p(psh), ctypes.c_float(0.0), int(masked),
                                   P, Cout, Cin, _p(dXb), _p(stb) if masked else None, 0, _p(wsb_t), wsb, st)
    assert rc == 0, L.pcl_last_error()
    rc = L.pcl_linear_bwd_pair_finish_f32(_p(wsb_t), wsb, P, Cout, Cin, _p(dWb), 0, None, 0, None, None, None, 0, None, None, None, None, None, None, st)
    assert rc == 0, L.pcl_last_error()
    torch.cuda.synchronize()
    assert torch.equal(dWa, dWb) and torch.equal(dXa, dXb)
    if masked:
        assert torch.equal(sta, stb)
    assert not torch.isnan(dWb).any() and not torch.isnan(dXb).any()
