import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.count)
for e in rows[:45]:
    print(f"{e.key[:70]:70s} n={e.count/3:6.1f}  cpu={e.cpu_time_total/3:8.1f}us  dev={e.device_time_total/3:8.1f}us")
