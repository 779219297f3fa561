"""Host enqueue time of one PointNet++ SSG step, split: forward / loss / backward / optimizer (no sync inside)."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from pointcloudlib_amd.affinity import pin_to_gpu_node; pin_to_gpu_node(0)
from pointcloudlib_amd import synth
from pointcloudlib_amd.networks.cls.pointnet2 import PointNet2_cls
from pointcloudlib_amd.train_utils import soft_cross_entropy_loss, make_sgd
torch.manual_seed(0)
net = PointNet2_cls().cuda().train()
opt = make_sgd(net.parameters(), lr=0.02, momentum=0.9)
x = torch.from_numpy(synth.gauss_ball(32, 1024, 1)).cuda(); f = torch.from_numpy(synth.unit_normals(32, 1024, 2)).cuda()
y = torch.from_numpy(synth.labels(32, 40, 3)).cuda()
T = [0.0] * 5
def step(rec):
    t0 = time.perf_counter()
    for p in net.parameters(): p.grad = None
    t1 = time.perf_counter()
    out = net(x, f)
    t2 = time.perf_counter()
    loss = soft_cross_entropy_loss(out, y)
    t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter()
    opt.step()
    t5 = time.perf_counter()
    if rec:
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): T[i] += d
for _ in range(10): step(False)
torch.cuda.synchronize()
n = 0
for rep in range(10):
    for _ in range(4): step(True); n += 1
    torch.cuda.synchronize()          # (drain: the launch queue must not back-pressure the host)
print("host us/step: zero_grad %.0f | forward %.0f | loss %.0f | backward %.0f | optimizer %.0f | total %.0f" % tuple([1e6 * t / n for t in T] + [1e6 * sum(T) / n]))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    out = net(x, f)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
