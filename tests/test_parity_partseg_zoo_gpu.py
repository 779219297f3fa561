"""GPU: the remaining part-segmentation callers (networks/seg/{dgcnn,pointnet,pointconv,pointcnn}_partseg.py) at the part-seg
driver's size -- B=16, N=2048 -- forward AND backward against their CPU restatements (oracle/cpu_partseg_zoo.py,
oracle/cpu_pointcnn.py) by the methodology of the BASELINE networks (oracle/parity.py): the restatement evaluated in fp32 and in
fp64 on the CPU, indices exact wherever they depend on coordinates only, features / logits elementwise against the fp64 value,
every parameter gradient by the fp64 yardstick with its absolute caps.  Nothing here compares the HIP path with itself.
Reference: /root/reference/networks/seg/dgcnn_partseg.py:36-128, pointnet_partseg.py:14-67, pointconv_partseg.py:9-63,
pointcnn_partseg.py:13-49 (loss: plain cross entropy over the 50 parts, train_partseg.py:116).
"""
import copy

import numpy as np
import pytest
import torch

from pointcloudlib_amd import synth

pytestmark = pytest.mark.gpu
B, N = 16, 2048


def _no_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return model


def _inputs():
    pts = synth.gauss_ball(B, N, 20247)
    onehot = torch.zeros(B, 16); onehot[torch.arange(B), torch.arange(B) % 16] = 1
    seg = torch.from_numpy(np.random.default_rng(6).integers(0, 50, (B, N)))
    return torch.from_numpy(pts), onehot, seg


_ce = torch.nn.functional.cross_entropy


def test_dgcnn_partseg_b16_n2048(oracle, dev):
    from oracle.cpu_partseg_zoo import DGCNNPartSegCPU
    from oracle.parity import Report
    from pointcloudlib_amd.networks.cls.dgcnn import knn_graph
    from pointcloudlib_amd.networks.seg.dgcnn_partseg import DGCNN_partseg
    torch.manual_seed(0)
    xyz, onehot, seg = _inputs()
    xt = xyz.transpose(1, 2).contiguous()
    net = _no_dropout(DGCNN_partseg(50).to(dev)).train()
    state = net.state_dict()
    r32, r64 = DGCNNPartSegCPU(state), DGCNNPartSegCPU(state, dtype=torch.float64)
    o32, aux = r32(xt, onehot, return_aux=True)
    lists = aux["lists"]
    o64, aux64 = r64(xt, onehot, lists=lists, return_aux=True)
    _ce(o32, seg).backward(); _ce(o64, seg).backward()
    xin, oh = xt.to(dev), onehot.to(dev)
    # kNN (k = 40) of every stage on the HIP network's OWN stage inputs against the oracle on the same bits: exact lists
    with torch.no_grad():
        _, stages = net(xin, oh, return_stages=True)
        own = []
        for s, t in enumerate((xin.transpose(1, 2).contiguous(),) + stages[:2]):
            got = knn_graph(t, net.knn).cpu().numpy()
            tt = np.ascontiguousarray(t.cpu().numpy().transpose(0, 2, 1))
            assert np.array_equal(got, oracle.knn(tt, tt, 40).transpose(0, 2, 1)), f"stage {s + 1}: kNN lists differ from the oracle"
            own.append(got)
    differ = [float((np.sort(own[s], -1) != np.sort(lists[s].numpy(), -1)).any(-1).mean()) for s in range(3)]
    assert differ[0] == 0.0
    for s in (1, 2):        # the HIP network's own feature-space lists vs the fp32 restatement's, as SETS (cf. test_parity_dgcnn_gpu.py)
        assert differ[s] <= 1e-3, f"stage {s + 1}: {differ[s]:.2e} of the points have a different own-feature kNN set"
    dev_lists = [l.to(dev).int().contiguous() for l in lists]
    out, stages = net(xin, oh, lists=dev_lists, return_stages=True)
    rep = Report(f"DGCNN part-seg B={B} N={N} k=40")
    for s in range(3):
        rep.feature(stages[s], aux["feats"][s], aux64["feats"][s], f"EdgeConv {s + 1} output")
    rep.feature(out, o32, o64, "logits [B,50,N]")
    loss = _ce(out, seg.to(dev))
    loss.backward()
    g_hip = {n: p.grad for n, p in net.named_parameters()}
    assert all(v is not None for v in g_hip.values())
    rep.grads(g_hip, {n: r32.grad(n) for n in g_hip}, {n: r64.grad(n) for n in g_hip})
    rep.check(abs(loss.item() - _ce(o64, seg).item()) <= 1e-5, "loss differs from the fp64 restatement")
    print("\n    fraction of points whose own-feature kNN SET differs from the fp32 restatement's, per stage: " + ", ".join(f"{d:.4f}" for d in differ))
    rep.finish()


def test_pointnet_partseg_b16_n2048(oracle, dev):
    from oracle.cpu_partseg_zoo import PointNetPartSegCPU
    from oracle.parity import Report
    from pointcloudlib_amd.networks.seg.pointnet_partseg import PointNet_partseg
    torch.manual_seed(1)
    xyz, onehot, seg = _inputs()
    xt = xyz.transpose(1, 2).contiguous()
    net = _no_dropout(PointNet_partseg().to(dev)).train()
    state = net.state_dict()
    r32, r64 = PointNetPartSegCPU(state), PointNetPartSegCPU(state, dtype=torch.float64)
    (o32, a32), (o64, a64) = r32(xt, onehot, return_aux=True), r64(xt, onehot, return_aux=True)
    with torch.no_grad():      # fp64 arithmetic, fp32 storage: the floor no fp32 summation scheme can beat (oracle/parity.py)
        o6s = PointNetPartSegCPU(state, dtype=torch.float64, storage="fp32")(xt, onehot)
    _ce(o32, seg).backward(); _ce(o64, seg).backward()
    out = net(xt.to(dev), onehot.to(dev))
    rep = Report(f"PointNet part-seg B={B} N={N}")
    rep.feature(out, o32, o64, "logits [B,50,N]", o6s)
    loss = _ce(out, seg.to(dev))
    loss.backward()
    g_hip = {n: p.grad for n, p in net.named_parameters()}
    assert all(v is not None for v in g_hip.values())
    rep.grads(g_hip, {n: r32.grad(n) for n in g_hip}, {n: r64.grad(n) for n in g_hip})
    rep.check(abs(loss.item() - _ce(o64, seg).item()) <= 1e-5, "loss differs from the fp64 restatement")
    rep.finish()


def test_pointconv_partseg_b16_n2048(oracle, dev, monkeypatch):
    from oracle.cpu_partseg_zoo import PointConvPartSegCPU
    from oracle.parity import Report
    from pointcloudlib_amd.misc import pointconv_utils as pu
    from pointcloudlib_amd.networks.seg.pointconv_partseg import PointConvDensity_partseg
    torch.manual_seed(2)
    xyz, onehot, seg = _inputs()
    net = _no_dropout(PointConvDensity_partseg().to(dev)).train()
    # the reference draws every FPS start with np.random.randint (misc/pointconv_utils.py:88): both sides take the same draws,
    # in call order sa0..sa3, in0..in3
    sizes = [N, 1024, 256, 64, 64, 256, 1024, N]
    rng = np.random.default_rng(3)
    start = [rng.integers(0, n, B).astype(np.int32) for n in sizes]
    calls = []
    real_fps = pu.farthest_point_sample

    def fps_with_given_start(x, npoint, start_idx=None):
        i = len(calls) % len(start)
        assert start_idx is None and x.shape[1] == sizes[i]
        calls.append(i)
        return real_fps(x, npoint, torch.from_numpy(start[i]).to(x.device))
    monkeypatch.setattr(pu, "farthest_point_sample", fps_with_given_start)
    state = net.state_dict()
    r32, r64 = PointConvPartSegCPU(state), PointConvPartSegCPU(state, dtype=torch.float64)
    (o32, a32), (o64, a64) = r32(xyz, start, return_aux=True), r64(xyz, start, return_aux=True)
    with torch.no_grad():
        o6s = PointConvPartSegCPU(state, dtype=torch.float64, storage="fp32")(xyz, start)
    tgt = seg.reshape(-1)
    lossf = lambda o, t: _ce(o.reshape(-1, 50), t)
    lossf(o32, tgt).backward(); lossf(o64, tgt).backward()
    out = net(xyz.to(dev), onehot.to(dev))
    assert calls == list(range(8))
    rep = Report(f"PointConv part-seg B={B} N={N}")
    rep.feature(out, o32, o64, "logits [B,N,50]", o6s)
    loss = lossf(out, tgt.to(dev))
    loss.backward()
    g_hip = {n: p.grad for n, p in net.named_parameters()}
    assert all(v is not None for v in g_hip.values())
    rep.grads(g_hip, {n: r32.grad(n) for n in g_hip}, {n: r64.grad(n) for n in g_hip})
    rep.check(abs(loss.item() - lossf(o64, tgt).item()) <= 1e-5, "loss differs from the fp64 restatement")
    rep.finish()


def test_pointcnn_partseg_b16_n2048(oracle, dev):
    from oracle import cpu_pointcnn as ref
    from oracle.parity import Report
    from pointcloudlib_amd.networks.seg.pointcnn_partseg import PointCNN_partseg
    torch.manual_seed(3)
    xyz, _, seg = _inputs()
    net = _no_dropout(PointCNN_partseg().to(dev)).train()
    r32, r64 = copy.deepcopy(net).cpu(), copy.deepcopy(net).cpu().double()
    o32 = ref.pointcnn_partseg(r32, xyz)
    o64 = ref.pointcnn_partseg(r64, xyz.double())
    _ce(o32, seg).backward(); _ce(o64, seg).backward()
    out = net(xyz.to(dev))
    rep = Report(f"PointCNN part-seg B={B} N={N}")
    rep.feature(out, o32, o64, "logits [B,50,N]")
    loss = _ce(out, seg.to(dev))
    loss.backward()
    g_hip = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
    g32 = {n: p.grad for n, p in r32.named_parameters() if p.grad is not None}
    g64 = {n: p.grad for n, p in r64.named_parameters() if p.grad is not None}
    assert set(g_hip) == set(g64) == set(g32)
    rep.grads(g_hip, g32, g64)
    rep.check(abs(loss.item() - _ce(o64, seg).item()) <= 1e-5, "loss differs from the fp64 restatement")
    rep.finish()
