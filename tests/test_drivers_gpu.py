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
