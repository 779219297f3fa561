"""The two training drivers end to end on synthetic data (they fall back to it when the datasets are absent): a couple of
epochs each, loss finite and decreasing on the training set, metrics in range.  Counterparts of
/root/reference/train_cls.py and train_partseg.py."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(script, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, script), *args], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def test_train_cls_driver():
    out = _run("train_cls.py", "--model", "pointnet2", "--epochs", "3", "--batch_size", "16", "--synthetic_items", "64",
               "--data_root", "/nonexistent")
    losses = [float(x) for x in re.findall(r"train loss ([\d.]+)", out)]
    assert len(losses) == 3 and all(l == l and l < 10 for l in losses) and min(losses[1:]) < losses[0], out


@pytest.mark.parametrize("model,extra", [("pointnet2", ()), ("pointnet2", ("--prefetch_sampling",)), ("dgcnn", ())])
def test_train_partseg_driver(model, extra):
    out = _run("train_partseg.py", "--model", model, "--epochs", "3", "--batch_size", "8", "--num_points", "512",
               "--synthetic_items", "32", "--data_root", "/nonexistent", *extra)
    tr = re.findall(r"Train \d+, loss: ([\d.]+), train acc: ([\d.]+), train avg acc: ([\d.]+), train iou: ([\d.]+)", out)
    te = re.findall(r"Test \d+, loss: ([\d.]+), test acc: ([\d.]+), test avg acc: ([\d.]+), test iou: ([\d.]+)", out)
    assert len(tr) == 3 and len(te) == 3, out
    losses = [float(t[0]) for t in tr]
    assert min(losses[1:]) < losses[0] < 6.0, out
    for row in tr + te:
        assert all(0.0 <= float(v) <= 1.0 for v in row[1:]), out


def test_lean_sgd_survives_load_state_dict(dev):
    """the cached momentum-buffer lists of _LeanFusedSGD must be dropped when load_state_dict replaces the buffers
    (a resume would otherwise keep updating the stale tensors and ignore the loaded momentum).  step, load, step against
    torch.optim.SGD(fused=True), bit for bit; same after the state is cleared behind the optimizer's back."""
    import copy
    import torch
    from pointcloudlib_amd.train_utils import make_sgd
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(33, 7, device=dev)), torch.nn.Parameter(torch.randn(5, device=dev))]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    a = make_sgd(ps, lr=0.02, momentum=0.9, weight_decay=1e-4)
    b = torch.optim.SGD(qs, lr=0.02, momentum=0.9, weight_decay=1e-4, fused=True)
    assert type(a).__name__ == "_LeanFusedSGD"

    def step():
        for p, q in zip(ps, qs):
            g = torch.randn_like(p)
            p.grad, q.grad = g.clone(), g.clone()
        a.step(); b.step()
        for p, q in zip(ps, qs):
            assert torch.equal(p, q)

    step(); step()                                        # second step runs on the cached lists
    saved_a, saved_b = copy.deepcopy(a.state_dict()), copy.deepcopy(b.state_dict())
    step(); step()
    a.load_state_dict(saved_a); b.load_state_dict(saved_b)
    step(); step()
    for p, q in zip(ps, qs):                              # and the saved state is the live one
        assert torch.equal(a.state[p]["momentum_buffer"], b.state[q]["momentum_buffer"])
    a.state.clear(); b.state.clear()                      # state re-created behind the optimizer's back
    step(); step()


@pytest.mark.parametrize("model", ["cls", "partseg_msg", "pointconv"])
def test_sampling_prefetch_equals_inline_sampling(dev, model):
    """The bench / train drivers produce the FPS + ball-query indices of batch t+1 on a side stream under the backward of
    batch t (SamplingPrefetch.precompute_sampling).  The handle's memory comes from the side stream's allocator pool and is
    consumed on the main stream WITHOUT record_stream (reuse is ordered by the wait_stream at the head of every
    precompute_sampling): 24 training steps over rotating batches must reproduce the inline-sampling run -- identical
    indices every step (an early reuse of a live index buffer would corrupt them), the first losses to rounding and the
    later ones to 2e-2 (the head's dX uses float atomics and max-pool winners flip on 1e-7 differences: two INLINE runs part
    the same way, by 1e-3 after twenty steps even at this tiny learning rate)."""
    import copy
    import torch
    from pointcloudlib_amd import synth
    from pointcloudlib_amd.train_utils import make_sgd, soft_cross_entropy_loss
    from pointcloudlib_amd.misc import head as _head
    torch.manual_seed(0)
    if model == "cls":
        from pointcloudlib_amd.networks.cls.pointnet2 import PointNet2_cls
        B, N = 16, 1024
        net_a = PointNet2_cls().to(dev).train()
        ys = [torch.from_numpy(synth.labels(B, 40, 30 + i)).to(dev) for i in range(3)]
        call = lambda net, x, f, y, s: soft_cross_entropy_loss(net(x, f, sampling=s), y)
    elif model == "pointconv":
        # (PointConv's handle also carries the kernel densities; FPS starts are random draws of torch's generator: both runs
        # draw them in the same order)
        from pointcloudlib_amd.networks.cls.pointconv import PointConvDensityClsSsg
        B, N = 8, 1024
        net_a = PointConvDensityClsSsg().to(dev).train()
        ys = [torch.from_numpy(synth.labels(B, 40, 30 + i)).to(dev) for i in range(3)]
        call = lambda net, x, f, y, s: soft_cross_entropy_loss(net(x.transpose(1, 2).contiguous(), sampling=s), y)
    else:
        from pointcloudlib_amd.networks.seg.pointnet2_partseg import PointNetMSG
        B, N = 4, 1024
        net_a = PointNetMSG(part_num=50).to(dev).train()
        ys = [torch.randint(0, 50, (B, N), device=dev) for i in range(3)]
        cls = torch.zeros(B, 16, device=dev); cls[:, 3] = 1
        call = lambda net, x, f, y, s: torch.nn.functional.cross_entropy(net(x, f, cls, sampling=s), y)
    net_b = copy.deepcopy(net_a)
    xs = [torch.from_numpy(synth.gauss_ball(B, N, 10 + i)).to(dev) for i in range(3)]
    fs = [torch.from_numpy(synth.unit_normals(B, N, 20 + i)).to(dev) for i in range(3)]
    side = "own"           # the network's private producer stream: the route without record_stream (pointnet2.sampling_stream)
    layout = (lambda t: t.transpose(1, 2).contiguous()) if model == "pointconv" else (lambda t: t)

    def run(net, prefetch):
        torch.manual_seed(1)                                  # the head's dropout masks: torch's CUDA generator (seed, offset), reset here
        _head._DROP_CALLS[0] = 0
        opt = make_sgd(net.parameters(), lr=1e-5, momentum=0.9)
        losses, digests, pending = [], [], {}
        for i in range(24):
            opt.zero_grad(set_to_none=True)
            s = pending.pop(i, None)
            if not prefetch:
                s = net.precompute_sampling(layout(xs[i % 3]))
            elif s is None:
                s = net.precompute_sampling(layout(xs[i % 3]), stream=side)
            loss = call(net, xs[i % 3], fs[i % 3], ys[i % 3], s)
            if prefetch:
                pending[i + 1] = net.precompute_sampling(layout(xs[(i + 1) % 3]), stream=side)
                assert pending[i + 1]["fed_from"] == torch.cuda.current_stream() and pending[i + 1]["owned"]
            loss.backward()
            opt.step()
            losses.append(loss.detach())
            # digest of every index tensor of the handle, taken on the consumer stream AFTER the whole step was enqueued
            d = torch.zeros((), dtype=torch.int64, device=dev)
            for new_xyz, idxs in s["levels"]:
                for ic in idxs:
                    for u in (ic or ()):
                        if u is not None and not u.is_floating_point():
                            d = d * 1000003 + u.to(torch.int64).sum()
            digests.append(d)
        torch.cuda.synchronize()
        return torch.stack(losses).cpu(), torch.stack(digests).cpu()

    la, da = run(net_a, False)
    lb, db = run(net_b, True)
    assert torch.equal(da, db), "sampling indices differ between the prefetched and the inline run"
    assert torch.allclose(la[:3], lb[:3], rtol=1e-5, atol=1e-6), (la - lb).abs()[:3]
    n_close = 8 if model == "pointconv" else 24       # (PointConv's steps are larger: two INLINE runs are 1e-1 apart after 24 of them)
    assert torch.allclose(la[:n_close], lb[:n_close], rtol=0, atol=2e-2), (la - lb).abs().max()

