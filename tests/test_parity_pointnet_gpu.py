"""GPU: PointNet classification (BASELINE configs[0], B=8 N=1024) against the CPU restatement of the reference's network
(oracle/cpu_pointnet.py) evaluated in fp32 and fp64 -- pooled features and logits by the 1e-5 rule, every parameter gradient by
the fp64 yardstick of oracle/parity.py (the methodology of the other four networks)."""
import numpy as np
import pytest
import torch

from pointcloudlib_amd import synth

pytestmark = pytest.mark.gpu


def test_pointnet_cls_b8_n1024(oracle, dev):
    from oracle.cpu_pointnet import PointNetClsCPU
    from oracle.parity import Report
    from pointcloudlib_amd.networks.cls.pointnet import PointNet
    from pointcloudlib_amd.train_utils import soft_cross_entropy_loss
    B, N = 8, 1024
    torch.manual_seed(0)
    pts, lab = synth.gauss_ball(B, N, 20241), synth.labels(B, 40, 21141)
    net = PointNet().to(dev).train()
    net.dp1.p = 0.0                                              # dropout off on both sides
    state = net.state_dict()
    r32, r64 = PointNetClsCPU(state), PointNetClsCPU(state, dtype=torch.float64)
    xin_c = torch.from_numpy(pts).transpose(1, 2).contiguous()
    o32, a32 = r32(xin_c, return_aux=True)
    o64, a64 = r64(xin_c, return_aux=True)
    soft_cross_entropy_loss(o32, torch.from_numpy(lab)).backward()
    soft_cross_entropy_loss(o64, torch.from_numpy(lab)).backward()
    xin = xin_c.to(dev)
    out = net(xin)
    soft_cross_entropy_loss(out, torch.from_numpy(lab).to(dev)).backward()
    rep = Report(f"PointNet cls B={B} N={N}")
    rep.feature(out, o32, o64, "logits")
    g_hip = {n: p.grad for n, p in net.named_parameters()}
    rep.grads(g_hip, {n: r32.grad(n) for n in g_hip}, {n: r64.grad(n) for n in g_hip})
    rep.finish()
