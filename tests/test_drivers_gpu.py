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
