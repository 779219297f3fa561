"""GPU: PointNet++ SSG cls (BASELINE configs[1]) against the CPU restatement AT ITS STATED SIZE -- B=32, N=1024 and N=4096.

Forward, level by level (reference composition networks/cls/pointnet2.py:33-62, :149-158):
  * FPS indices, sampled centres and ball-query neighbour lists bit-exact at EVERY level;
  * pooled features and logits elementwise within ``atol = rtol = 1e-5`` (north_star: "within 1e-5 fp32 for features";
    numpy.allclose semantics, |a-b| <= atol + rtol*|b|) of the restatement evaluated in fp64 -- the exact value of the
    reference's arithmetic, which every fp32 implementation of it (Jittor's, PyTorch-CPU's, ours) only approximates --
    and within that bound PLUS the fp32 restatement's own measured distance from the fp64 value of the fp32 restatement
    (PyTorch-CPU fp32 is 2-2.5x further from the fp64 value than the HIP path is: 1.6e-5 vs 7.5e-6 at SA2, B=32).
Backward (the reference step is forward+backward, train_cls.py:54-75): gradients of all 1 472 552 parameters and of the
input features against the restatement.  Two fp32 pipelines cannot agree to 1e-5 on gradients: a max-pool winner that
flips between two rows whose pre-BatchNorm outputs agree to 1 ulp moves a whole gradient row, and BatchNorm backward
divides by the batch std three levels deep.  The yardstick is therefore the SAME restatement in fp64: the HIP gradient
must be as close to the fp64 truth as the fp32 restatement itself is (factor GRAD_SLACK, oracle/parity.py), per tensor, in
relative L2 and in max norm -- and both numbers are printed (measured here: the HIP gradients are 4-15x CLOSER to the
fp64 value than PyTorch-CPU fp32's).
"""
import numpy as np
import pytest
import torch

from pointcloudlib_amd import synth

pytestmark = pytest.mark.gpu



def _no_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return model


@pytest.mark.parametrize("N", [1024, 4096])
def test_pointnet2_ssg_b32_levels_and_gradients(oracle, dev, N):
    from oracle.cpu_model import PointNet2ClsCPU
    from oracle.parity import Report
    from pointcloudlib_amd.misc import ops
    from pointcloudlib_amd.networks.cls.pointnet2 import PointNet2_cls
    from pointcloudlib_amd.train_utils import soft_cross_entropy_loss
    B = 32
    torch.manual_seed(0)
    pts, nrm, lab = synth.gauss_ball(B, N, 20242), synth.unit_normals(B, N, 20742), synth.labels(B, 40, 21142)
    net = _no_dropout(PointNet2_cls().to(dev)).train()
    S = ops.optimal_block(B)
    state = net.state_dict()
    ref32 = PointNet2ClsCPU(state, tie_stride=S).train()
    ref64 = PointNet2ClsCPU(state, tie_stride=S, dtype=torch.float64).train()
    x, f, y = torch.from_numpy(pts).to(dev), torch.from_numpy(nrm).to(dev), torch.from_numpy(lab).to(dev)

    # ---- CPU side: forward with per-level tensors, then backward (fp32 restatement and fp64 truth)
    f32 = torch.from_numpy(nrm).clone().requires_grad_(True)
    f64 = torch.from_numpy(nrm).double().requires_grad_(True)
    logits32, aux = ref32(torch.from_numpy(pts), f32, return_aux=True)
    logits64, aux64 = ref64(torch.from_numpy(pts), f64, return_aux=True)
    soft_cross_entropy_loss(logits32, torch.from_numpy(lab)).backward()
    soft_cross_entropy_loss(logits64, torch.from_numpy(lab)).backward()

    # ---- forward, level by level
    report = Report(f"PointNet++ SSG cls B=32 N={N}")
    with torch.no_grad():
        cur_xyz, cur_f = x, f
        for lvl, mod in enumerate(net.pointnet_modules):
            if mod.n_points is not None:
                fidx, fxyz = ops.furthest_point_sample(cur_xyz, mod.n_points)
                assert np.array_equal(fidx.cpu().numpy(), aux[lvl]["fps_idx"]), f"SA{lvl + 1}: FPS indices differ"
                samp = mod.sample(cur_xyz)
                assert torch.equal(samp[0], fxyz), f"SA{lvl + 1}: module centres != sampler centres"
                want_xyz = np.take_along_axis(cur_xyz.cpu().numpy(), aux[lvl]["fps_idx"][:, :, None].astype(np.int64), axis=1)
                assert np.array_equal(samp[0].cpu().numpy(), want_xyz), f"SA{lvl + 1}: sampled centres differ"
                assert np.array_equal(samp[1][0][0].cpu().numpy(), aux[lvl]["bq_idx"]), f"SA{lvl + 1}: ball-query lists differ"
            else:
                samp = None
            new_xyz, cur_f = mod(cur_xyz, cur_f, samp)
            report.feature(cur_f, aux[lvl]["feat"], aux64[lvl]["feat"], f"SA{lvl + 1} pooled features")
            if new_xyz is not None:
                cur_xyz = new_xyz
    # ---- whole step on the production path (normals carry no gradient: the first conv is folded inline)
    out = net(x, f)
    report.feature(out, logits32, logits64, "logits")
    loss = soft_cross_entropy_loss(out, y)
    loss.backward()
    g_hip = {n: p.grad.detach().cpu().double() for n, p in net.named_parameters()}
    assert len(g_hip) == sum(1 for _ in net.parameters()) and sum(v.numel() for v in g_hip.values()) == 1472552
    # ---- second pass with a gradient for the input features (the first conv then runs as a per-point GEMM)
    net.zero_grad(set_to_none=True)
    fg = f.clone().requires_grad_(True)
    soft_cross_entropy_loss(net(x, fg), y).backward()
    g_hip2 = {n: p.grad.detach().cpu().double() for n, p in net.named_parameters()}
    g_hip["<input features>"] = fg.grad.detach().cpu().double()

    def cpu_grads(ref, fgrad):
        d = {n: ref.p[ref.keys[n]].grad for n in g_hip if n != "<input features>"}
        d["<input features>"] = fgrad
        return d

    g32, g64 = cpu_grads(ref32, f32.grad), cpu_grads(ref64, f64.grad)
    report.grads(g_hip, g32, g64)
    report.grads(g_hip2, g32, g64, tag="(feature-gradient pass) ")      # same bars for every parameter
    lerr = abs(loss.item() - soft_cross_entropy_loss(logits64, torch.from_numpy(lab)).item())
    report.check(lerr <= 1e-5, f"loss differs from the fp64 restatement by {lerr:.2e}")
    report.finish()
