"""How much of the host time of a PointNet++ step is ctypes calls, torch.empty, and the rest (wrappers with perf_counter)."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from pointcloudlib_amd.affinity import pin_to_gpu_node; pin_to_gpu_node(0)
from pointcloudlib_amd import synth, _lib
from pointcloudlib_amd.misc import mlp_hip
from pointcloudlib_amd.networks.cls.pointnet2 import PointNet2_cls
from pointcloudlib_amd.train_utils import soft_cross_entropy_loss, make_sgd
torch.manual_seed(0)
net = PointNet2_cls().cuda().train()
opt = make_sgd(net.parameters(), lr=0.02, momentum=0.9)
x = torch.from_numpy(synth.gauss_ball(32, 1024, 1)).cuda(); f = torch.from_numpy(synth.unit_normals(32, 1024, 2)).cuda()
y = torch.from_numpy(synth.labels(32, 40, 3)).cuda()
def step():
    for p in net.parameters(): p.grad = None
    soft_cross_entropy_loss(net(x, f), y).backward(); opt.step()
for _ in range(10): step()
torch.cuda.synchronize()
acc = {"call": 0.0, "ncall": 0, "empty": 0.0, "nempty": 0}
orig_call, orig_empty = _lib.call, torch.empty
def call(*a, **k):
    t = time.perf_counter(); r = orig_call(*a, **k); acc["call"] += time.perf_counter() - t; acc["ncall"] += 1; return r
def empty(*a, **k):
    t = time.perf_counter(); r = orig_empty(*a, **k); acc["empty"] += time.perf_counter() - t; acc["nempty"] += 1; return r
_lib.call = call; torch.empty = empty
import pointcloudlib_amd.misc.ops as ops, pointcloudlib_amd.misc.head as head
n = 40
t0 = time.perf_counter()
for i in range(n):
    step()
    if i % 4 == 3: torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("per step: total host+sync %.0f us | _lib.call %.0f us in %d calls (%.1f us each) | torch.empty %.0f us in %d calls (%.1f us each)" % (
    1e6 * tot / n, 1e6 * acc["call"] / n, acc["ncall"] / n, 1e6 * acc["call"] / max(1, acc["ncall"]),
    1e6 * acc["empty"] / n, acc["nempty"] / n, 1e6 * acc["empty"] / max(1, acc["nempty"])))
