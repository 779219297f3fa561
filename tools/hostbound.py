import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from pointcloudlib_amd import synth
from pointcloudlib_amd.networks.cls.pointnet2 import PointNet2_cls
from pointcloudlib_amd.train_utils import soft_cross_entropy_loss
torch.manual_seed(0)
net = PointNet2_cls().cuda().train()
opt = torch.optim.SGD(net.parameters(), lr=0.02, momentum=0.9)
x = torch.from_numpy(synth.gauss_ball(32, 1024, 1)).cuda(); f = torch.from_numpy(synth.unit_normals(32, 1024, 2)).cuda()
y = torch.from_numpy(synth.labels(32, 40, 3)).cuda()
def step():
    opt.zero_grad(set_to_none=True)
    loss = soft_cross_entropy_loss(net(x, f), y); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0)/20:.3f} ms/step   total {1e3*(t2-t0)/20:.3f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
