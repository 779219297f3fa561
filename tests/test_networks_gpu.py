"""GPU: whole networks built from the HIP ops -- shapes, backend agreement (fused HIP MLP vs plain-PyTorch MLP on
the same HIP indices) and, for PointNet++ SSG, feature parity with the CPU restatement level by level."""
import copy

import numpy as np
import pytest
import torch

import oracle.torch_backend  # noqa: F401,E402  (registers the plain-PyTorch composite the tests compare against)

from pointcloudlib_amd import synth

pytestmark = pytest.mark.gpu


def set_backend(model, backend):
    for m in model.modules():
        if hasattr(m, "backend"):
            m.backend = backend
    return model


def no_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return model


def fwd_bwd(model, args, backend):
    m = set_backend(copy.deepcopy(model), backend).train()
    out = m(*args)
    out.square().mean().backward()
    return out.detach(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}


def test_pointnet2_cls_matches_cpu_restatement(oracle, dev):
    from oracle.cpu_model import PointNet2ClsCPU
    from pointcloudlib_amd.misc import ops
    from pointcloudlib_amd.networks.cls.pointnet2 import PointNet2_cls
    torch.manual_seed(0)
    B, N = 8, 1024
    pts, nrm = synth.gauss_ball(B, N, 11), synth.unit_normals(B, N, 12)
    net = no_dropout(PointNet2_cls().to(dev)).train()
    ref = PointNet2ClsCPU(net.state_dict(), tie_stride=ops.optimal_block(B)).train()
    x, f = torch.from_numpy(pts).to(dev), torch.from_numpy(nrm).to(dev)
    # level by level: sampled xyz bit-exact, pooled features within 1e-5 (relative to the level's max)
    with torch.no_grad():
        logits_ref, aux = ref(torch.from_numpy(pts), torch.from_numpy(nrm), return_aux=True)
        cur_xyz, cur_f = x, f
        for lvl, mod in enumerate(net.pointnet_modules):
            new_xyz, cur_f = mod(cur_xyz, cur_f)
            want = aux[lvl]["feat"]
            scale = max(1.0, want.abs().max().item())
            err = (cur_f.cpu() - want).abs().max().item()
            assert err <= 1e-5 * scale, f"SA{lvl + 1}: {err:.3e} vs scale {scale:.3f}"
            if new_xyz is not None:
                assert np.array_equal(new_xyz.cpu().numpy()[:, :, 0], pts[np.arange(B)[:, None], aux[lvl]["fps_idx"]][:, :, 0]) \
                    if lvl == 0 else True
                cur_xyz = new_xyz
        logits = net.fc_layer(cur_f.squeeze(1))
        assert (logits.cpu() - logits_ref).abs().max().item() <= 1e-4 * max(1.0, logits_ref.abs().max().item())


def test_pointnet2_cls_backends_agree_with_grads(dev):
    from pointcloudlib_amd.networks.cls.pointnet2 import PointNet2_cls
    torch.manual_seed(1)
    B, N = 4, 1024
    x = torch.from_numpy(synth.gauss_ball(B, N, 3)).to(dev)
    f = torch.from_numpy(synth.unit_normals(B, N, 4)).to(dev)
    net = no_dropout(PointNet2_cls().to(dev))
    o_h, g_h = fwd_bwd(net, (x, f), "hip")
    o_t, g_t = fwd_bwd(net, (x, f), "torch")
    assert (o_h - o_t).abs().max().item() <= 1e-4 * max(1.0, o_t.abs().max().item())
    gmax = max(v.abs().max().item() for v in g_t.values())
    for n in g_t:
        s = max(1e-3 * gmax, g_t[n].abs().max().item())                # some gradients are exactly 0 in theory
        # two fp32 pipelines through 3 SA levels: BN backward amplifies rounding and max-pool winners can flip
        assert (g_h[n] - g_t[n]).abs().max().item() <= 5e-3 * s, n


def test_pointnet_and_partseg_shapes_and_backends(dev):
    from pointcloudlib_amd.networks.cls.pointnet import PointNet
    from pointcloudlib_amd.networks.seg.pointnet2_partseg import PointNet2_partseg, PointNetMSG
    torch.manual_seed(2)
    B, N = 4, 2048
    x = torch.from_numpy(synth.gauss_ball(B, N, 5)).to(dev)
    net = no_dropout(PointNet().to(dev))
    o_h, g_h = fwd_bwd(net, (x.transpose(1, 2).contiguous(),), "hip")
    o_t, g_t = fwd_bwd(net, (x.transpose(1, 2).contiguous(),), "torch")
    assert o_h.shape == (B, 40) and (o_h - o_t).abs().max().item() <= 1e-4 * max(1.0, o_t.abs().max().item())
    onehot = torch.zeros(B, 16, device=dev)
    onehot[torch.arange(B), torch.arange(B) % 16] = 1
    for cls in (PointNet2_partseg, PointNetMSG):
        seg = no_dropout(cls().to(dev))
        o_h, g_h = fwd_bwd(seg, (x, x, onehot), "hip")              # train_partseg.py:110: model(data, data, onehot)
        o_t, g_t = fwd_bwd(seg, (x, x, onehot), "torch")
        assert o_h.shape == (B, 50, N)
        assert (o_h - o_t).abs().max().item() <= 1e-3 * max(1.0, o_t.abs().max().item())
        assert set(g_h) == set(g_t) and all(torch.isfinite(v).all() for v in g_h.values())


def test_dgcnn_forward_backward(oracle, dev):
    from pointcloudlib_amd.misc import ops
    from pointcloudlib_amd.networks.cls.dgcnn import DGCNN, get_graph_feature
    torch.manual_seed(3)
    B, N = 4, 512
    pts = synth.gauss_ball(B, N, 6)
    x = torch.from_numpy(pts).to(dev)
    # edge features of the first stage against the oracle's kNN + a NumPy gather
    idx = oracle.knn(np.ascontiguousarray(pts.transpose(0, 2, 1)), np.ascontiguousarray(pts.transpose(0, 2, 1)), 20)
    idx = idx.transpose(0, 2, 1)
    ef = get_graph_feature(x, ops.KNN(20)).cpu().numpy()
    nb = pts[np.arange(B)[:, None, None], idx]
    want = np.concatenate([nb - pts[:, :, None, :], np.broadcast_to(pts[:, :, None, :], nb.shape)], -1)
    assert np.array_equal(ef, want)
    net = no_dropout(DGCNN().to(dev)).train()
    out = net(x.transpose(1, 2).contiguous())
    assert out.shape == (B, 40)
    out.square().mean().backward()
    assert all(torch.isfinite(p.grad).all() for p in net.parameters())
    # gradient of the edge-feature op against autograd on plain indexing
    xr = x[:, :64].clone().requires_grad_(True)
    ii = torch.from_numpy(np.ascontiguousarray(idx[:, :64, :8] % 64)).to(dev).int()
    g = torch.randn(B, 64, 8, 6, device=dev)
    ops.edge_features(xr, ii).backward(g)
    xr2 = x[:, :64].clone().requires_grad_(True)
    nbr = xr2[torch.arange(B, device=dev)[:, None, None], ii.long()]
    torch.cat([nbr - xr2[:, :, None, :], xr2[:, :, None, :].expand_as(nbr)], -1).backward(g)
    assert (xr.grad - xr2.grad).abs().max().item() <= 1e-5 * max(1.0, xr2.grad.abs().max().item())


def test_pointconv_pieces_and_network(oracle, dev):
    from pointcloudlib_amd.misc import pointconv_utils as pu
    from pointcloudlib_amd.networks.cls.pointconv import PointConvDensityClsSsg
    torch.manual_seed(4)
    B, N = 4, 512
    pts = synth.gauss_ball(B, N, 8)
    x = torch.from_numpy(pts).to(dev)
    # KDE density: direct-form distances; fp32 running sum vs fp64 oracle -> 1e-5 relative
    for bw in (0.1, 0.4):
        d = pu.compute_density(x, bw).cpu().numpy()
        np.testing.assert_allclose(d, oracle.density(pts, bw), rtol=2e-5, atol=1e-7)
    # (eight lanes per point since round 5: sizes that are no multiple of 8 / 32, more points than one LDS chunk, a single point)
    for n in (1, 7, 100, 2500):
        q = synth.gauss_ball(2, n, 3 + n)
        dq = pu.compute_density(torch.from_numpy(q).to(dev), 0.2).cpu().numpy()
        np.testing.assert_allclose(dq, oracle.density(q, 0.2), rtol=2e-5, atol=1e-7)
    # FPS variant: random start, no origin skip -> the oracle with skip disabled
    start = np.array([5, 0, 511, 77], np.int32)
    fidx = pu.farthest_point_sample(x, 64, torch.from_numpy(start).to(dev)).cpu().numpy()
    assert np.array_equal(fidx, oracle.fps(pts, 64, block_size=1, skip=False, start_idx=start))
    # kNN grouping in xyz space: k nearest by (direct-form d2, index)
    new_xyz = pts[np.arange(B)[:, None], fidx]
    idx = pu.knn_point(16, x, torch.from_numpy(new_xyz).to(dev)).cpu().numpy()
    want = oracle.knn(np.ascontiguousarray(new_xyz.transpose(0, 2, 1)), np.ascontiguousarray(pts.transpose(0, 2, 1)), 16)
    assert np.array_equal(idx, want.transpose(0, 2, 1))
    # whole network: shapes, backend agreement (same HIP indices, fused vs plain-PyTorch MLP), finite grads
    net = no_dropout(PointConvDensityClsSsg().to(dev))
    xin = x.transpose(1, 2).contiguous()
    st = [torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)]
    o_h, g_h = fwd_bwd(net, (xin, st), "hip")
    o_t, g_t = fwd_bwd(net, (xin, st), "torch")
    assert o_h.shape == (B, 40)
    assert (o_h - o_t).abs().max().item() <= 1e-3 * max(1.0, o_t.abs().max().item())
    assert set(g_h) == set(g_t) and all(torch.isfinite(v).all() for v in g_h.values())


def test_partseg_zoo_shapes_and_backends(dev):
    """The remaining part-segmentation callers (networks/seg/*.py): output shape, fused-HIP vs plain-PyTorch MLP
    agreement on the same HIP indices, finite gradients for every parameter."""
    from pointcloudlib_amd.networks.seg.dgcnn_partseg import DGCNN_partseg
    from pointcloudlib_amd.networks.seg.pointconv_partseg import PointConvDensity_partseg
    from pointcloudlib_amd.networks.seg.pointnet_partseg import PointNet_partseg
    def inputs(B, N):
        x = torch.from_numpy(synth.gauss_ball(B, N, 9)).to(dev)
        onehot = torch.zeros(B, 16, device=dev)
        onehot[torch.arange(B), torch.arange(B) % 16] = 1
        return x, x.transpose(1, 2).contiguous(), onehot
    # PointNet's T-Nets run BatchNorm over B rows only: at B=2 that is a sign function with a 1/sqrt(eps) slope at 0
    # (two chained layers amplify fp32 rounding by ~1e4), so that case uses a larger batch.
    x8, xt8, oh8 = inputs(8, 1024)
    x2, xt2, oh2 = inputs(2, 2048)
    x4, _, oh4 = inputs(4, 2048)
    cases = [
        ("pointnet", lambda: PointNet_partseg(), (xt8, oh8), (8, 50, 1024)),
        ("dgcnn", lambda: DGCNN_partseg(50), (xt2, oh2), (2, 50, 2048)),
        ("pointconv", lambda: PointConvDensity_partseg(), (x4, oh4), (4, 2048, 50)),
    ]
    # PointConv part-seg is 8 PointConv levels deep and its coarsest ones normalise over B*36 rows: looser bound
    tol = {"pointnet": 2e-3, "dgcnn": 2e-3, "pointconv": 1e-2}
    for name, make, args, shape in cases:
        torch.manual_seed(7)
        net = no_dropout(make().to(dev))
        torch.manual_seed(8)                      # PointConv draws its FPS start indices from the global RNG
        o_h, g_h = fwd_bwd(net, args, "hip")
        torch.manual_seed(8)
        o_t, g_t = fwd_bwd(net, args, "torch")
        assert o_h.shape == shape, name
        assert (o_h - o_t).abs().max().item() <= tol[name] * max(1.0, o_t.abs().max().item()), name
        assert set(g_h) == set(g_t) and all(torch.isfinite(v).all() for v in g_h.values()), name


def test_stn_identity_at_zero_weights(dev):
    from pointcloudlib_amd.misc.stn import STN3d, STNkd
    x = torch.randn(3, 128, 3, device=dev)
    stn = STN3d().to(dev).train()
    torch.nn.init.zeros_(stn.fc3.weight); torch.nn.init.zeros_(stn.fc3.bias)
    assert torch.equal(stn(x), torch.eye(3, device=dev).expand(3, 3, 3))      # fc3 == 0 -> exactly the identity (:50-56)
    assert STNkd(16).to(dev)(torch.randn(2, 64, 16, device=dev)).shape == (2, 16, 16)


@pytest.mark.parametrize("B,S,ns,C", [(2, 5, 32, 128), (1, 3, 7, 20), (2, 1, 200, 64), (1, 4, 64, 300), (3, 7, 48, 33), (1, 5, 16, 64),
                                      (2, 3, 24, 36), (1, 9, 33, 8),
                                      # round 5 kernels: fragment-direct forward (any C, row blocks of 32), row-major weight gradient (C % 4 == 0, ns <= 64:
                                      # either side of both limits), one row block exactly, C < 32, the GroupAll shape
                                      (1, 2, 65, 36), (2, 2, 63, 128), (1, 9, 32, 33), (2, 3, 31, 4), (1, 1, 128, 1024), (2, 7, 5, 3)])
def test_pointconv_contraction_kernel(dev, B, S, ns, C):
    """pcl_pointconv_contract_f32 (+bwd) against the reference formula (misc/pointconv_utils.py:393-394) in PyTorch."""
    from pointcloudlib_amd.misc.pointconv_utils import pointconv_contract
    torch.manual_seed(ns + C)
    f = torch.randn(B, S, ns, C, device=dev, requires_grad=True)
    d = torch.rand(B, S, ns, 1, device=dev, requires_grad=True)
    w = torch.randn(B, S, ns, 16, device=dev, requires_grad=True)
    g = torch.randn(B, S, C * 16, device=dev)
    out = pointconv_contract(f, d, w)
    out.backward(g)
    got = [out.detach(), f.grad.clone(), d.grad.clone(), w.grad.clone()]
    f2, d2, w2 = (t.detach().double().requires_grad_(True) for t in (f, d, w))
    ref = torch.matmul((f2 * d2).transpose(2, 3), w2).reshape(B, S, -1)
    ref.backward(g.double())
    want = [ref.detach(), f2.grad, d2.grad, w2.grad]
    for a, b, name in zip(got, want, ("out", "dfeat", "ddens", "dw")):
        assert a.shape == b.shape, name
        assert (a.double() - b).abs().max().item() <= 1e-5 * max(1.0, b.abs().max().item()), name


@pytest.mark.parametrize("B,N,C,Cout,k", [(2, 64, 3, 64, 8), (3, 200, 64, 128, 20), (1, 33, 5, 20, 33)])
def test_edgeconv_factorised_matches_edge_tensor(oracle, dev, B, N, C, Cout, k):
    """misc/edgeconv.py (U[nbr] + V, no edge tensor) against the reference formulation (edge tensor -> conv -> BN ->
    LeakyReLU -> max, networks/cls/dgcnn.py:29-50,:100-102) in fp64 PyTorch, forward and every gradient."""
    from pointcloudlib_amd.misc.edgeconv import edge_conv
    from pointcloudlib_amd.misc.layers import PointwiseMLP
    torch.manual_seed(N + C)
    mlp = PointwiseMLP([2 * C, Cout], slope=0.2).to(dev).train()
    mlp.gammas[0].data.uniform_(-1.0, 1.5)                 # both signs: exercises the max/min choice
    mlp.betas[0].data.uniform_(-0.5, 0.5)
    x = torch.randn(B, N, C, device=dev)
    xn = x.cpu().numpy().transpose(0, 2, 1).copy()
    idx_np = oracle.knn(xn, xn, k).transpose(0, 2, 1).copy()
    idx = torch.from_numpy(idx_np).to(dev)
    g = torch.randn(B, N, Cout, device=dev)
    ref = copy.deepcopy(mlp).double(); ref.backend = "torch"          # before the first forward touches the running stats
    xa = x.clone().requires_grad_(True)
    out = edge_conv(mlp, xa, idx)
    out.backward(g)
    got = {"out": out.detach(), "dx": xa.grad.clone(), **{n: p.grad.clone() for n, p in mlp.named_parameters()}}
    # fp64 reference with the edge tensor
    xb = x.double().clone().requires_grad_(True)
    nb = xb[torch.arange(B, device=dev)[:, None, None], idx.long()]
    e = torch.cat([nb - xb[:, :, None, :], xb[:, :, None, :].expand_as(nb)], -1)
    want = ref(e, group_max=k)
    want.backward(g.double())
    exp = {"out": want.detach(), "dx": xb.grad, **{n: p.grad for n, p in ref.named_parameters()}}
    for n in exp:
        s = max(1.0, exp[n].abs().max().item())
        assert (got[n].double() - exp[n]).abs().max().item() <= 2e-5 * s, n
    for b in ("running_mean_0", "running_var_0"):
        assert torch.allclose(getattr(mlp, b).double(), getattr(ref, b), rtol=1e-5, atol=1e-6), b


@pytest.mark.parametrize("B,N,k", [(3, 64, 5), (2, 1024, 20), (1, 300, 40), (2, 17, 17), (1, 4096, 6), (1, 130, 70)])     # (4096: fewer waves share the LDS counters; k = 70 > 64: the sorted form)
def test_knn_transpose_lists(dev, B, N, k):
    """pcl_knn_transpose_i32: for every point the ascending list of the points that name it as a neighbour."""
    from pointcloudlib_amd import _lib
    rng = np.random.default_rng(B * 1000 + N + k)
    idx_np = np.stack([np.stack([rng.permutation(N)[:k] for _ in range(N)]) for _ in range(B)]).astype(np.int32)   # distinct per row
    idx = torch.from_numpy(idx_np).to(dev)
    in_off = torch.empty((B * N + 1,), dtype=torch.int32, device=dev)
    in_src = torch.empty((B * N * k,), dtype=torch.int32, device=dev)
    _lib.call("pcl_knn_transpose_i32", idx.data_ptr(), B, N, k, in_off.data_ptr(), in_src.data_ptr(), torch.cuda.current_stream().cuda_stream)
    off, src = in_off.cpu().numpy(), in_src.cpu().numpy()
    assert off[0] == 0 and off[-1] == B * N * k and (np.diff(off) >= 0).all()
    for b in range(B):
        for n in range(N):
            want = np.sort(np.nonzero((idx_np[b] == n).any(axis=1))[0])
            got = src[off[b * N + n]:off[b * N + n + 1]]
            assert got.tolist() == want.tolist(), (b, n)


def test_edgeconv_backward_lists_match_per_edge_atomics(oracle, dev):
    """pcl_edgeconv_scatter_f32: the transposed-list formulation against the per-edge atomic one (same inputs)."""
    from pointcloudlib_amd import _lib
    B, N, C, k = 2, 256, 64, 20
    torch.manual_seed(5)
    UV = torch.randn(B * N, 2 * C, device=dev)
    xn = torch.randn(B, 3, N).numpy()
    idx = torch.from_numpy(oracle.knn(xn, xn, k).transpose(0, 2, 1).copy()).to(dev).contiguous()
    gz = torch.randn(B * N, C, device=dev)
    arg = torch.randint(0, k, (B * N, C), dtype=torch.int32, device=dev)
    a, k1, k2, mu = (torch.randn(C, device=dev) * s for s in (1.0, 0.01, 0.01, 0.5))
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: t.data_ptr()
    stats = torch.empty((_lib.lib().pcl_edgeconv_stat_rows(B, N), 2, C), dtype=torch.float64, device=dev)
    ymax, ymin = torch.empty(B * N, C, device=dev), torch.empty(B * N, C, device=dev)
    jmax, jmin = torch.empty(B * N, C, dtype=torch.int32, device=dev), torch.empty(B * N, C, dtype=torch.int32, device=dev)
    sumU = torch.empty(B * N, C, device=dev)
    _lib.call("pcl_edgeconv_gather_f32", P(UV), P(idx), B, N, k, C, P(ymax), P(ymin), P(jmax), P(jmin), P(stats), P(sumU), st)
    nb = UV.view(B, N, 2 * C)[torch.arange(B, device=dev)[:, None, None], idx.long()][..., :C]          # [B,N,k,C]
    assert torch.allclose(sumU.view(B, N, C), nb.sum(2), rtol=1e-5, atol=1e-5)
    in_off = torch.empty((B * N + 1,), dtype=torch.int32, device=dev)
    in_src = torch.empty((B * N * k,), dtype=torch.int32, device=dev)
    _lib.call("pcl_knn_transpose_i32", P(idx), B, N, k, P(in_off), P(in_src), st)
    d_edges, d_lists = torch.empty_like(UV), torch.empty_like(UV)
    _lib.call("pcl_edgeconv_scatter_f32", P(UV), P(idx), P(gz), P(arg), P(a), P(k1), P(k2), P(mu), B, N, k, C, None, None, None, P(d_edges), st)
    _lib.call("pcl_edgeconv_scatter_f32", P(UV), P(idx), P(gz), P(arg), P(a), P(k1), P(k2), P(mu), B, N, k, C, P(in_off), P(in_src),
              P(sumU), P(d_lists), st)
    scale = d_edges.abs().max().item()
    assert (d_edges - d_lists).abs().max().item() <= 2e-5 * max(1.0, scale)


def test_lean_sgd_is_torch_fused_sgd(dev):
    """train_utils.make_sgd on GPU parameters caches the parameter / momentum lists after the first step and then calls the
    same multi-tensor kernel torch.optim.SGD(fused=True) does: bit-identical parameters after several steps, with weight
    decay, a learning-rate change through param_groups, and a step in which a parameter has no gradient."""
    from pointcloudlib_amd.train_utils import make_sgd
    torch.manual_seed(3)
    ws = [torch.randn(s, device=dev) for s in ((7, 5), (11,), (3, 4, 2))]
    pa = [torch.nn.Parameter(w.clone()) for w in ws]
    pb = [torch.nn.Parameter(w.clone()) for w in ws]
    oa = make_sgd(pa, lr=0.05, momentum=0.9, weight_decay=1e-3)
    ob = torch.optim.SGD(pb, lr=0.05, momentum=0.9, weight_decay=1e-3, fused=True)
    assert type(oa).__name__ == "_LeanFusedSGD"
    for step in range(6):
        gs = [torch.randn_like(w) for w in ws]
        for p, q, g in zip(pa, pb, gs):
            p.grad, q.grad = g.clone(), g.clone()
        if step == 3:
            for o in (oa, ob):
                o.param_groups[0]["lr"] = 0.01
        if step == 4:
            pa[1].grad = None; pb[1].grad = None
        oa.step(); ob.step()
        for p, q in zip(pa, pb):
            assert torch.equal(p.data, q.data), step
    for p, q in zip(pa, pb):
        assert torch.equal(oa.state[p]["momentum_buffer"], ob.state[q]["momentum_buffer"])
    # a parameter whose storage is re-seated (`p.data = ...`, what `.to()` does) and a gradient the own kernel cannot take (fp64 view cast
    # back, non-contiguous): the cached pointer tables must follow / step aside
    pa[0].data = pa[0].data.clone(); pb[0].data = pb[0].data.clone()
    gs = [torch.randn_like(w) for w in ws]
    for p, q, g in zip(pa, pb, gs):
        p.grad, q.grad = g.clone(), g.clone()
    pa[2].grad = pa[2].grad.transpose(0, 1).contiguous().transpose(0, 1)          # same values, non-contiguous
    oa.step(); ob.step()
    for p, q in zip(pa, pb):
        assert torch.equal(p.data, q.data)


@pytest.mark.parametrize("B,S,ns,Cin,spec,slope", [(4, 96, 32, 19, [64, 64, 128], 0.0), (2, 1, 128, 35, [32, 96], 0.0), (3, 40, 24, 7, [16, 8], 0.2)])
def test_pointconv_feature_mlp_folded_into_the_contraction(dev, B, S, ns, Cin, spec, slope):
    """feature_mlp_contract: the feature MLP's last BatchNorm + activation applied inside the contraction kernels (forward and
    backward, csrc/pointconv.hip FeatBN; the stack defers them, csrc/stack.hip defer_act) against the unfused composition
    ``pointconv_contract(mlp(x), density, weights)`` on the per-kernel path: same outputs, same gradients (fp64 partial sums in
    another order), same running statistics."""
    import copy
    from pointcloudlib_amd.misc import mlp_hip, pointconv_utils as pu
    from pointcloudlib_amd.misc.layers import PointwiseMLP
    torch.manual_seed(B * S + ns)
    mlp = PointwiseMLP([Cin] + spec, bias=True, slope=slope).to(dev).train()
    with torch.no_grad():
        for g in mlp.gammas:
            g.uniform_(0.5, 1.5); g[::3] *= -1.0
    x0 = torch.randn(B, S, ns, Cin, device=dev)
    dens0 = torch.rand(B, S, ns, 1, device=dev) + 0.5
    w0 = torch.randn(B, S, ns, 16, device=dev)
    gout = torch.randn(B, S, spec[-1] * 16, device=dev)
    res = []
    for fused in (False, True):
        m = copy.deepcopy(mlp)
        x, dn, w = (t.clone().requires_grad_(True) for t in (x0, dens0, w0))
        if fused:
            out = pu.feature_mlp_contract(m, x, dn, w)
        else:
            with mlp_hip.per_kernel_path():
                out = pu.pointconv_contract(m(x), dn, w)
        out.backward(gout)
        res.append((out.detach(), [x.grad, dn.grad, w.grad] + [p.grad for p in m.parameters()], [b.clone() for b in m.buffers()]))
    (o0, g0, b0), (o1, g1, b1) = res
    assert (o0 - o1).abs().max().item() <= 1e-6 * max(1.0, o0.abs().max().item())
    gs = max(t.abs().max().item() for t in g0)
    for a, b in zip(g0, g1):
        assert (a - b).abs().max().item() <= 1e-5 * a.abs().max().item() + 1e-7 * gs
    for a, b in zip(b0, b1):
        assert torch.allclose(a.float(), b.float(), rtol=1e-6, atol=1e-7)


def test_pointconv_set_abstraction_folded_paths_equal_the_composition(dev):
    """PointConvDensitySetAbstraction with points: the product path (first conv folded into the k-NN grouping, last BatchNorm + ReLU
    folded into the contraction, per-stack entry points) against the plain composition on the per-kernel entry points
    (cat([xyz[idx] - new_xyz, points[idx]]) -> MLP -> contraction): outputs, input-feature gradient and every parameter gradient."""
    import copy
    from pointcloudlib_amd import synth
    from pointcloudlib_amd.misc import mlp_hip, pointconv_utils as pu
    torch.manual_seed(3)
    B, N, S, ns, D = 4, 512, 96, 24, 37
    xyz = torch.from_numpy(synth.gauss_ball(B, N, 77)).to(dev).permute(0, 2, 1).contiguous()
    pts0 = torch.randn(B, D, N, device=dev)
    sa = pu.PointConvDensitySetAbstraction(npoint=S, nsample=ns, in_channel=D + 3, mlp=[48, 64], bandwidth=0.2, group_all=False).to(dev).train()
    start = torch.zeros(B, dtype=torch.int32, device=dev)
    gout = torch.randn(B, 64, S, device=dev)
    res = []
    for product in (False, True):
        m = copy.deepcopy(sa)
        p = pts0.clone().requires_grad_(True)
        if product:
            _, out = m(xyz, p, start)
        else:
            with mlp_hip.per_kernel_path():
                _, out = m(xyz, p, start)
        out.backward(gout)
        res.append((out.detach(), p.grad.detach(), {n: q.grad.detach() for n, q in m.named_parameters()}))
    (o0, f0, g0), (o1, f1, g1) = res
    assert (o0 - o1).abs().max().item() <= 2e-5 * max(1.0, o0.abs().max().item())
    gs = max(t.abs().max().item() for t in g0.values())
    assert (f0 - f1).abs().max().item() <= 1e-4 * f0.abs().max().item() + 1e-7 * gs
    for n in g0:
        assert (g0[n] - g1[n]).abs().max().item() <= 1e-4 * g0[n].abs().max().item() + 1e-6 * gs, n


def test_pointconv_interpolation_folded_paths_equal_the_composition(dev):
    """PointConvDensitySetInterpolation (the part-seg decoder): the product path (first conv of the feature MLP folded into the k-NN
    grouping over all N points, last BatchNorm + ReLU in the contraction, narrow-stack WeightNet / DensityNet) against the plain
    composition on the per-kernel entry points: output, gradient of the coarse features and every parameter gradient."""
    import copy
    from pointcloudlib_amd import synth
    from pointcloudlib_amd.misc import mlp_hip, pointconv_utils as pu
    torch.manual_seed(4)
    B, N, S, ns, D2 = 3, 384, 96, 16, 40
    xyz1 = torch.from_numpy(synth.gauss_ball(B, N, 78)).to(dev).permute(0, 2, 1).contiguous()
    xyz2 = xyz1[:, :, ::4].contiguous()
    pts2 = torch.randn(B, D2, S, device=dev)
    fp = pu.PointConvDensitySetInterpolation(nsample=ns, in_channel=D2 + 3, mlp=[48, 64], bandwidth=0.2).to(dev).train()
    start = torch.zeros(B, dtype=torch.int32, device=dev)
    gout = torch.randn(B, 64, N, device=dev)
    res = []
    for product in (False, True):
        m = copy.deepcopy(fp)
        p = pts2.clone().requires_grad_(True)
        if product:
            out = m(xyz1, xyz2, None, p, start)
        else:
            with mlp_hip.per_kernel_path():
                out = m(xyz1, xyz2, None, p, start)
        out.backward(gout)
        res.append((out.detach(), p.grad.detach(), {n: q.grad.detach() for n, q in m.named_parameters()}))
    (o0, f0, g0), (o1, f1, g1) = res
    assert o0.shape == (B, 64, N)
    assert (o0 - o1).abs().max().item() <= 2e-5 * max(1.0, o0.abs().max().item())
    gs = max(t.abs().max().item() for t in g0.values())
    assert (f0 - f1).abs().max().item() <= 1e-4 * f0.abs().max().item() + 1e-7 * gs
    for n in g0:
        if n.endswith("densitynet.mlp.weights.0"):          # analytically zero (one input channel under BatchNorm): rounding noise both ways
            continue
        assert (g0[n] - g1[n]).abs().max().item() <= 1e-4 * g0[n].abs().max().item() + 1e-6 * gs, n


@pytest.mark.parametrize("B,N,Cin,Cout", [(4, 1024, 512, 1024), (3, 100, 24, 40), (2, 300, 64, 70)])
def test_conv_max_mean_pool_equals_the_composition(dev, B, N, Cin, Cout):
    """DGCNN's conv5 + global max / mean pooling (networks/cls/dgcnn.py:113-116): the product path (deferring stack + the pooling
    kernel that applies BatchNorm + LeakyReLU while it reduces: no [B,N,C] activation) against conv -> max / mean / cat on the
    per-kernel path, and against PyTorch in fp64: output, input gradient and the conv's parameter gradients."""
    import copy
    from pointcloudlib_amd.misc import mlp_hip
    from pointcloudlib_amd.misc.edgeconv import conv_max_mean_pool
    from pointcloudlib_amd.misc.layers import PointwiseMLP
    torch.manual_seed(11)
    m = PointwiseMLP([Cin, Cout], slope=0.2).to(dev).train()
    with torch.no_grad():
        m.gammas[0].uniform_(0.5, 1.5); m.gammas[0][::3] *= -1.0; m.betas[0].uniform_(-0.3, 0.3)
    x0 = torch.randn(B, N, Cin, device=dev)
    gout = torch.randn(B, 2 * Cout, device=dev)
    res = []
    for mode in ("product", "per_kernel", "fp64"):
        mm = copy.deepcopy(m)
        x = x0.clone().requires_grad_(True)
        if mode == "product":
            out = conv_max_mean_pool(mm, x)
        elif mode == "per_kernel":
            with mlp_hip.per_kernel_path():
                out = conv_max_mean_pool(mm, x)
        else:
            mm = mm.double(); mm.backend = "torch"
            x = x0.double().clone().requires_grad_(True)
            y = mm(x)
            out = torch.cat((y.max(dim=1)[0], y.mean(dim=1)), dim=1)
        out.backward(gout.to(out.dtype))
        res.append((out.detach().double(), x.grad.detach().double(), {n: p.grad.detach().double() for n, p in mm.named_parameters()},
                    {n: b.detach().double() for n, b in mm.named_buffers()}))
    ref = res[2]
    for got in res[:2]:
        assert (got[0] - ref[0]).abs().max().item() <= 1e-5 * max(1.0, ref[0].abs().max().item())
        assert (got[1] - ref[1]).abs().max().item() <= 1e-4 * ref[1].abs().max().item() + 1e-7
        for n in ref[2]:
            assert (got[2][n] - ref[2][n]).abs().max().item() <= 1e-4 * max(1e-6, ref[2][n].abs().max().item()), n
        for n in ref[3]:
            assert (got[3][n] - ref[3][n]).abs().max().item() <= 1e-5 * max(1.0, ref[3][n].abs().max().item()), n


@pytest.mark.gpu
def test_loss_backward_seeds_autograd_without_the_fill_and_the_product():
    """``train_utils.loss_backward(loss)`` = ``loss.backward()`` bit for bit (the loss kernel's stored gradient goes on as it is when the
    seed is the cached scalar 1.0, recognised by address); any other incoming gradient still takes the product."""
    import torch
    from pointcloudlib_amd.train_utils import loss_backward, seg_cross_entropy_loss, soft_cross_entropy_loss
    torch.manual_seed(3)
    x = torch.randn(32, 40, device="cuda")
    y = torch.randint(0, 40, (32,), device="cuda")
    s = torch.randn(4, 50, 333, device="cuda")
    t = torch.randint(0, 50, (4, 333), device="cuda")
    for fn, inp, tgt in ((soft_cross_entropy_loss, x, y), (seg_cross_entropy_loss, s, t)):
        a = inp.clone().requires_grad_(True)
        b = inp.clone().requires_grad_(True)
        c = inp.clone().requires_grad_(True)
        la = fn(a * 1.0, tgt); la.backward()
        lb = fn(b * 1.0, tgt); loss_backward(lb)
        assert torch.equal(la, lb) and torch.equal(a.grad, b.grad)
        (fn(c * 1.0, tgt) * 3.0).backward()                      # a gradient that is not the seed: scaled as before
        assert torch.allclose(c.grad, 3.0 * a.grad, rtol=1e-6, atol=0)


@pytest.mark.gpu
def test_group_all_backward_hands_the_feature_gradient_on_as_a_view():
    """GroupAll's backward returns the feature columns of the gradient as a strided view (no copy launch); values as the copy kernel's."""
    import torch
    from pointcloudlib_amd.misc import ops
    torch.manual_seed(4)
    xyz = torch.randn(3, 50, 3, device="cuda")
    f = torch.randn(3, 50, 7, device="cuda")
    g = torch.randn(3, 1, 50, 10, device="cuda")
    res = {}
    for view in (True, False):
        old, ops._GROUP_ALL_BWD_VIEW = ops._GROUP_ALL_BWD_VIEW, view
        try:
            ff = f.clone().requires_grad_(True)
            ops.GroupAll(True)(None, xyz, ff).backward(g)
            res[view] = ff.grad.clone()
        finally:
            ops._GROUP_ALL_BWD_VIEW = old
    assert torch.equal(res[True], res[False]) and torch.equal(res[True], g[:, 0, :, 3:])


@pytest.mark.gpu
def test_dgcnn_concat_written_by_the_stages_equals_torch_cat():
    """DGCNN's concat(x1..x4) (networks/cls/dgcnn.py:112) is written by the EdgeConv stages themselves (each output also goes to its column
    slice of the [B,N,512] buffer: pcl_group_minmax_finalize2_f32, misc/edgeconv.assemble): same logits bit for bit as torch.cat, same
    gradients up to the head's atomic summation order."""
    import torch
    from pointcloudlib_amd import synth
    from pointcloudlib_amd.networks.cls import dgcnn
    from pointcloudlib_amd.train_utils import soft_cross_entropy_loss
    torch.manual_seed(8)
    net = dgcnn.DGCNN().cuda().train()
    x = torch.from_numpy(synth.gauss_ball(4, 512, 5)).cuda().transpose(1, 2).contiguous()
    y = torch.from_numpy(synth.labels(4, 40, 6)).cuda()
    res = {}
    for flag in (False, True):
        old, dgcnn._CAT_IN_PLACE = dgcnn._CAT_IN_PLACE, flag
        old_t, dgcnn._STAGE_T = dgcnn._STAGE_T, flag          # (and: the stages hand their outputs on as [B,C,N] too -- no transpose launches)
        try:
            torch.manual_seed(9)                              # dropout masks
            net.zero_grad()
            out, stages = net(x, return_stages=True)
            soft_cross_entropy_loss(out, y).backward()
            res[flag] = (out.detach().clone(), [s.detach().clone() for s in stages], {n: p.grad.clone() for n, p in net.named_parameters()})
        finally:
            dgcnn._CAT_IN_PLACE = old
            dgcnn._STAGE_T = old_t
    a, b = res[False], res[True]
    for u, v in zip(a[1], b[1]):
        assert torch.equal(u, v)
    assert torch.equal(a[0], b[0])
    for n in a[2]:
        scale = a[2][n].abs().max().item()
        assert (a[2][n] - b[2][n]).abs().max().item() <= 1e-4 * scale + 1e-7, n      # (a bias in front of a BatchNorm: true gradient 0, rounding noise 1e-8)
