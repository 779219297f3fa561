"""GPU: PointCNN built on the HIP KNN / FPS / gathers / fused MLP kernels against the NCHW CPU restatement of
misc/layers.py (oracle/cpu_pointcnn.py): region indices bit-exact, features within 1e-4 of the level's scale."""
import copy

import numpy as np
import pytest
import torch

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
