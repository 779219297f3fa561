"""GPU: PointCNN built on the HIP KNN / FPS / gathers / fused MLP kernels against the NCHW CPU restatement of
misc/layers.py (oracle/cpu_pointcnn.py): region indices bit-exact, features within 1e-4 of the level's scale."""
import copy

import numpy as np
import pytest
import torch

import oracle.torch_backend  # noqa: F401,E402  (registers the plain-PyTorch composite the tests compare against)

from pointcloudlib_amd import synth

pytestmark = pytest.mark.gpu


def no_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return model


def cpu_copy(model):
    return copy.deepcopy(model).cpu()


@pytest.mark.parametrize("C_in,C_out,K,D,P", [(3, 48, 8, 1, -1), (48, 96, 12, 2, 96), (0, 32, 8, 3, 64)])
def test_pointcnn_stage_matches_restatement(oracle, dev, C_in, C_out, K, D, P):
    from oracle import cpu_pointcnn as ref
    from pointcloudlib_amd.misc.pointcnn import RandPointCNN
    torch.manual_seed(K + D)
    B, N = 4, 256
    pts = synth.gauss_ball(B, N, 30 + K)
    fts = np.random.default_rng(K).standard_normal((B, N, C_in)).astype(np.float32) if C_in else None
    mod = RandPointCNN(C_in, C_out, 3, K, D, P).to(dev).train()
    x = torch.from_numpy(pts).to(dev)
    f = torch.from_numpy(fts).to(dev).requires_grad_(True) if C_in else None
    rep, out = mod((x, f))
    out.square().mean().backward()

    cm = cpu_copy(mod)
    cm.zero_grad()
    fc = torch.from_numpy(fts).requires_grad_(True) if C_in else None
    rep_ref, want = ref.rand_pointcnn(cm, torch.from_numpy(pts), fc, oracle.optimal_block(B))
    want.square().mean().backward()
    assert np.array_equal(rep.cpu().numpy(), rep_ref.numpy())                       # sampled representatives bit-exact
    # dilated region indices bit-exact
    got_idx = mod.pointcnn.region_indices(rep, x).cpu().numpy()
    _, want_idx = ref.pointcnn(cm.pointcnn, rep_ref, torch.from_numpy(pts), fc.detach() if C_in else None, return_idx=True)
    assert np.array_equal(got_idx, want_idx)
    scale = max(1.0, want.abs().max().item())
    assert (out.detach().cpu() - want.detach()).abs().max().item() <= 1e-4 * scale
    gp = dict(mod.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in cm.parameters() if p.grad is not None)
    for n, p in cm.named_parameters():
        if p.grad is None:
            continue
        s = max(1e-3 * gmax, p.grad.abs().max().item())
        assert (gp[n].grad.cpu() - p.grad).abs().max().item() <= 5e-3 * s, n
    if C_in:
        s = max(1e-6, fc.grad.abs().max().item())
        assert (f.grad.cpu() - fc.grad).abs().max().item() <= 5e-3 * s


def test_pointcnn_cls_network(oracle, dev):
    from oracle import cpu_pointcnn as ref
    from pointcloudlib_amd.networks.cls.pointcnn import PointCNNcls
    torch.manual_seed(5)
    B, N = 4, 1024
    pts = synth.gauss_ball(B, N, 41)
    net = no_dropout(PointCNNcls().to(dev)).train()
    x = torch.from_numpy(pts).to(dev)
    out = net(x)
    assert out.shape == (B, 40)
    out.square().mean().backward()
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    with torch.no_grad():
        want = ref.pointcnn_cls(cpu_copy(net), torch.from_numpy(pts))
    assert (out.detach().cpu() - want).abs().max().item() <= 1e-3 * max(1.0, want.abs().max().item())


def test_pointcnn_cls_b32_n1024_parity(oracle, dev):
    """PointCNN cls at the classification drivers' size (B=32, N=1024), forward AND backward, by the methodology of the four
    BASELINE networks (oracle/parity.py): the NCHW restatement of misc/layers.py evaluated in fp32 and in fp64 on the CPU; logits
    elementwise within 1e-5 of the fp64 value, every parameter gradient by the fp64 yardstick with its absolute caps.
    (Sampling / region indices depend on the coordinates only: identical in all three pipelines, checked per stage above.)"""
    from oracle import cpu_pointcnn as ref
    from oracle.parity import Report
    from pointcloudlib_amd.networks.cls.pointcnn import PointCNNcls
    from pointcloudlib_amd.train_utils import soft_cross_entropy_loss
    torch.manual_seed(5)
    B, N = 32, 1024
    pts, lab = synth.gauss_ball(B, N, 20246), torch.from_numpy(synth.labels(B, 40, 21146))
    net = no_dropout(PointCNNcls().to(dev)).train()
    r32, r64 = cpu_copy(net), cpu_copy(net).double()
    x_c = torch.from_numpy(pts)
    o32 = ref.pointcnn_cls(r32, x_c)
    o64 = ref.pointcnn_cls(r64, x_c.double())
    soft_cross_entropy_loss(o32, lab).backward()
    soft_cross_entropy_loss(o64, lab).backward()
    out = net(x_c.to(dev))
    loss = soft_cross_entropy_loss(out, lab.to(dev))
    loss.backward()
    rep = Report(f"PointCNN cls B={B} N={N}")
    rep.feature(out, o32, o64, "logits")
    g_hip = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
    g32 = {n: p.grad for n, p in r32.named_parameters() if p.grad is not None}
    g64 = {n: p.grad for n, p in r64.named_parameters() if p.grad is not None}
    assert set(g_hip) == set(g64) == set(g32)
    rep.grads(g_hip, g32, g64)
    rep.check(abs(loss.item() - soft_cross_entropy_loss(o64, lab).item()) <= 1e-5, "loss differs from the fp64 restatement")
    rep.finish()


def test_pointcnn_partseg_network(oracle, dev):
    from oracle import cpu_pointcnn as ref
    from pointcloudlib_amd.networks.seg.pointcnn_partseg import PointCNN_partseg
    torch.manual_seed(6)
    B, N = 2, 2048
    pts = synth.gauss_ball(B, N, 43)
    net = PointCNN_partseg().to(dev).train()
    x = torch.from_numpy(pts).to(dev)
    out = net(x)
    assert out.shape == (B, 50, N)
    out.square().mean().backward()
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    with torch.no_grad():
        want = ref.pointcnn_partseg(cpu_copy(net), torch.from_numpy(pts))
    assert (out.detach().cpu() - want).abs().max().item() <= 1e-3 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("K,C1,C2,dm,R", [(8, 12, 24, 16, (3, 50)), (12, 24, 48, 2, (2, 77)), (16, 48, 96, 2, (2, 40)),
                                          (16, 96, 192, 2, (2, 33)), (8, 8, 0, 4, (5, 13)), (16, 300, 0, 1, (1, 9))])
def test_xconv_core_matches_fp64_composite(dev, K, C1, C2, dm, R):
    """csrc/xconv.hip (X @ [F1|F2], depthwise (1,K) conv, one kernel; misc/layers.py:505 + :151) forward and backward against
    the same arithmetic in fp64 PyTorch: every classifier stage's shape, the 288-channel stage (two channel windows in
    backward), no second feature tensor, dm = 1.  Tolerance: 1e-5 of the tensor's scale (fp32 sums of <= 16 + 16 terms)."""
    from pointcloudlib_amd.misc.pointcnn import xconv_core
    torch.manual_seed(K + C1)
    B, P = R
    C = C1 + C2
    X = torch.randn(B, P, K, K, dtype=torch.float64)
    F1 = torch.randn(B, P, K, C1, dtype=torch.float64)
    F2 = torch.randn(B, P, K, C2, dtype=torch.float64) if C2 else None
    wd = torch.randn(C, dm, K, dtype=torch.float64) * 0.3
    bias = torch.randn(C * dm, dtype=torch.float64)
    gD = torch.randn(B, P, C * dm, dtype=torch.float64)

    def leaves(dt, device):
        return [t.detach().to(device=device, dtype=dt).requires_grad_(True) if t is not None else None for t in (X, F1, F2, wd, bias)]

    r = leaves(torch.float64, "cpu")
    F = r[1] if r[2] is None else torch.cat((r[1], r[2]), dim=-1)
    want = torch.einsum("bpkc,cjk->bpcj", torch.matmul(r[0], F), r[3]).reshape(B, P, C * dm) + r[4]
    want.backward(gD)
    h = leaves(torch.float32, dev)
    got = xconv_core(*h)
    got.backward(gD.float().to(dev))
    pairs = [("D", got, want)] + [(n, a.grad, b.grad) for n, a, b in zip(("dX", "dF1", "dF2", "dwd", "dbias"), h, r) if a is not None]
    for name, a, b in pairs:
        scale = max(1.0, b.abs().max().item())
        assert (a.detach().double().cpu() - b.detach()).abs().max().item() <= 1e-5 * scale, name
