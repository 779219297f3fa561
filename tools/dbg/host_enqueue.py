"""Host time to ENQUEUE one PointNet++ step with the stream empty behind it: k steps back to back after a sync, for k = 1, 2, 4
(if the figure grows with k the launch queue throttles the host and the larger k measure GPU time, not host time)."""
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
params = list(net.parameters())
side = "own"
pending = {}
def step(i, prefetch):
    for p in params: p.grad = None
    samp = pending.pop(i, None) if prefetch else None
    out = net(x, f, sampling=samp)
    if prefetch:
        pending[i + 1] = net.precompute_sampling(x, stream=side)
    soft_cross_entropy_loss(out, y).backward(); opt.step()
import gc
for prefetch in (False, True):
    it = 0
    for _ in range(20): step(it, prefetch); it += 1
    torch.cuda.synchronize()
    gc.collect(); gc.disable()
    for k in (1, 2, 4):
        ts = []
        for rep in range(30):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k): step(it, prefetch); it += 1
            ts.append((time.perf_counter() - t0) / k)
        ts.sort()
        print(f"prefetch={prefetch} k={k}: host enqueue per step median {1e3 * ts[len(ts) // 2]:.3f} ms  min {1e3 * ts[0]:.3f} ms")
    gc.enable()
    pending.clear()
