"""CPU: the channel-last X-conv blocks (plain-PyTorch backend, named explicitly) against the NCHW restatement of
misc/layers.py in oracle/cpu_pointcnn.py -- pins the weight-index mapping of the (1,K) convs, the concat order and
the activation/BatchNorm order before any GPU is involved."""
import copy

import pytest
import torch

import oracle.torch_backend  # noqa: F401,E402  (registers the plain-PyTorch composite the tests compare against)


def _torch_backend(model):
    for m in model.modules():
        if hasattr(m, "backend"):
            m.backend = "torch"
    return model


@pytest.mark.parametrize("C_in,C_out,K,dm", [(6, 16, 4, 3), (0, 8, 3, 4), (12, 24, 5, 2)])
def test_xconv_layout_matches_nchw_restatement(C_in, C_out, K, dm):
    from oracle import cpu_pointcnn as ref
    from pointcloudlib_amd.misc.pointcnn import XConv
    torch.manual_seed(C_in + K)
    B, P = 2, 7
    mod = _torch_backend(XConv(C_in, C_out, 3, K, P, C_mid=C_out // 4, depth_multiplier=dm)).train()
    for p in mod.parameters():                      # non-trivial BN affine parameters
        if p.dim() == 1:
            p.data.uniform_(0.5, 1.5)
    rep = torch.randn(B, P, 3)
    pts = rep[:, :, None, :] + 0.3 * torch.randn(B, P, K, 3)
    fts = torch.randn(B, P, K, C_in, requires_grad=True) if C_in else None
    got = mod((rep, pts, fts))
    got.square().sum().backward()
    g_got = {n: p.grad.clone() for n, p in mod.named_parameters()}
    gf_got = fts.grad.clone() if C_in else None

    mod2 = copy.deepcopy(mod)
    mod2.zero_grad()
    fts2 = fts.detach().clone().requires_grad_(True) if C_in else None
    want = ref.xconv(mod2, rep, pts, fts2)
    want.square().sum().backward()
    assert got.shape == (B, P, C_out)
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-5)
    for n, p in mod2.named_parameters():
        if p.grad is None:                          # bias in front of a BatchNorm: zero gradient on both sides
            assert g_got[n].abs().max() < 1e-4, n
            continue
        s = max(1.0, p.grad.abs().max().item())
        assert (g_got[n] - p.grad).abs().max().item() <= 2e-4 * s, n
    if C_in:
        assert torch.allclose(gf_got, fts2.grad, rtol=1e-3, atol=1e-5)


def test_sepconv_and_conv_weight_mapping():
    from oracle import cpu_pointcnn as ref
    from pointcloudlib_amd.misc.pointcnn import Conv, SepConv
    torch.manual_seed(3)
    B, P, K, C = 2, 5, 4, 6
    x = torch.randn(B, P, K, C)
    conv = _torch_backend(Conv(C, 10, (1, K))).train()
    assert torch.allclose(conv(x), ref.conv_1xk(conv, x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).squeeze(2),
                          rtol=1e-4, atol=1e-5)
    sep = _torch_backend(SepConv(C, 9, (1, K), depth_multiplier=3)).train()
    assert torch.allclose(sep(x), ref.sep_conv(sep, x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).squeeze(2),
                          rtol=1e-4, atol=1e-5)
    # no-BN variants carry a bias
    conv = _torch_backend(Conv(C, 10, (1, K), with_bn=False)).train()
    assert conv.linear.biases is not None
    assert torch.allclose(conv(x), ref.conv_1xk(conv, x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).squeeze(2),
                          rtol=1e-4, atol=1e-5)
