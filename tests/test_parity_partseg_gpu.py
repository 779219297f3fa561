"""GPU: PointNet++ part segmentation (BASELINE configs[3]: MSG, ShapeNet-sized clouds, B=16, N=2048 xyz+normal) against
oracle/cpu_partseg.py at the stated size; the SSG variant rides along.

Encoder: FPS indices and every scale's ball-query lists exact, pooled features within 1e-5 of the fp64 value; decoder:
3-NN indices exact and interpolation weights to 1e-6, feature-propagation outputs and the [B,50,N] logits within 1e-5;
gradients of every parameter by the fp64 yardstick (oracle/parity.py).  Reference: networks/seg/pointnet2_partseg.py:110-176.
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


@pytest.mark.parametrize("variant", ["msg", "ssg"])
def test_pointnet2_partseg_b16_n2048(oracle, dev, variant):
    from oracle.cpu_partseg import PointNet2PartSegCPU
    from oracle.parity import Report
    from pointcloudlib_amd.misc import ops
    from pointcloudlib_amd.networks.seg.pointnet2_partseg import PointNet2_partseg, PointNetMSG
    B, N = 16, 2048
    torch.manual_seed(0)
    pts, nrm = synth.gauss_ball(B, N, 20244), synth.unit_normals(B, N, 20744)
    seg = np.random.default_rng(5).integers(0, 50, (B, N))
    onehot = torch.zeros(B, 16); onehot[torch.arange(B), torch.arange(B) % 16] = 1
    cls, spec = (PointNetMSG, PointNet2PartSegCPU.MSG) if variant == "msg" else (PointNet2_partseg, PointNet2PartSegCPU.SSG)
    net = _no_dropout(cls().to(dev)).train()
    S = ops.optimal_block(B)
    state = net.state_dict()
    r32 = PointNet2PartSegCPU(state, spec, tie_stride=S)
    r64 = PointNet2PartSegCPU(state, spec, tie_stride=S, dtype=torch.float64)
    xyz_c, nrm_c = torch.from_numpy(pts), torch.from_numpy(nrm)
    o32, a32 = r32(xyz_c, nrm_c, onehot, return_aux=True)
    o64, a64 = r64(xyz_c, nrm_c, onehot, return_aux=True)
    with torch.no_grad():      # fp64 arithmetic, fp32 storage: the floor no fp32 summation scheme can beat (oracle/parity.py)
        o6s, a6s = PointNet2PartSegCPU(state, spec, tie_stride=S, dtype=torch.float64, storage="fp32")(xyz_c, nrm_c, onehot, return_aux=True)
    tgt = torch.from_numpy(seg)
    lossf = torch.nn.functional.cross_entropy                                     # train_partseg.py: CE over the 50 parts
    lossf(o32, tgt).backward(); lossf(o64, tgt).backward()

    xyz, f, oh = xyz_c.to(dev), nrm_c.to(dev), onehot.to(dev)
    report = Report(f"PointNet++ part-seg {variant.upper()} B={B} N={N}")
    with torch.no_grad():
        cur_xyz, cur_f, lv = xyz, f, []
        for i, mod in enumerate(net.pointnet_modules):
            if mod.n_points is not None:
                fidx, _ = ops.furthest_point_sample(cur_xyz, mod.n_points)
                assert np.array_equal(fidx.cpu().numpy(), a32["sa"][i]["fps_idx"]), f"SA{i + 1}: FPS indices differ"
                samp = mod.sample(cur_xyz)
                for j, ic in enumerate(samp[1]):
                    assert np.array_equal(ic[0].cpu().numpy(), a32["sa"][i]["bq_idx"][j]), f"SA{i + 1} scale {j}: ball-query lists differ"
            else:
                samp = None
            new_xyz, cur_f = mod(cur_xyz, cur_f, samp)
            report.feature(cur_f, a32["sa"][i]["feat"], a64["sa"][i]["feat"], f"SA{i + 1} pooled features", a6s["sa"][i]["feat"])
            lv.append((cur_xyz, new_xyz, cur_f))
            if new_xyz is not None:
                cur_xyz = new_xyz
        # decoder 3-NN searches (fp2: l1 <- l2, fp1: input <- l1) against the oracle
        for (q, s, rec) in ((lv[1][0], lv[1][1], a32["fp"][0]), (xyz, lv[0][1], a32["fp"][1])):
            idx3, w3 = ops.three_nn(q, s)
            assert np.array_equal(idx3.cpu().numpy(), rec["three_nn"]), "3-NN indices differ"
            np.testing.assert_allclose(w3.cpu().numpy(), rec["weights"], rtol=1e-6, atol=1e-7)
        # decoder, module by module (networks/seg/pointnet2_partseg.py:168-173)
        l1_xyz, l1_f, l2_xyz, l2_f, l3_f = lv[0][1], lv[0][2], lv[1][1], lv[1][2], lv[2][2]
        d2 = net.fp3(l2_xyz, torch.zeros((B, 1, 3), device=dev), l2_f, l3_f)
        report.feature(d2, a32["decoder"][0], a64["decoder"][0], "fp3 output", a6s["decoder"][0])
        d1 = net.fp2(l1_xyz, l2_xyz, l1_f, d2)
        report.feature(d1, a32["decoder"][1], a64["decoder"][1], "fp2 output", a6s["decoder"][1])
        d0 = net.fp1(xyz, l1_xyz, torch.cat([oh.view(B, 1, 16).expand(B, N, 16), xyz, f], 2), d1)
        report.feature(d0, a32["decoder"][2], a64["decoder"][2], "fp1 output", a6s["decoder"][2])
    out = net(xyz, f, oh)
    report.feature(out, o32, o64, "logits [B,50,N]", o6s)
    loss = lossf(out, tgt.to(dev))
    loss.backward()
    g_hip = {n: p.grad for n, p in net.named_parameters()}
    assert all(v is not None for v in g_hip.values())
    report.grads(g_hip, {n: r32.grad(n) for n in g_hip}, {n: r64.grad(n) for n in g_hip})
    report.check(abs(loss.item() - lossf(o64, tgt).item()) <= 1e-5, 'loss differs from the fp64 restatement')
    report.finish()
