"""cProfile of whole PointNet++ steps with autograd's worker thread disabled (backward runs on the profiled thread)."""
import os, sys, time, torch, cProfile, pstats
sys.path.insert(0, os.getcwd())
from pointcloudlib_amd.affinity import pin_to_gpu_node; pin_to_gpu_node(0)
from pointcloudlib_amd import synth
from pointcloudlib_amd.networks.cls.pointnet2 import PointNet2_cls
from pointcloudlib_amd.train_utils import soft_cross_entropy_loss, make_sgd
torch.manual_seed(0)
torch.autograd.set_multithreading_enabled(False)
net = PointNet2_cls().cuda().train()
opt = make_sgd(net.parameters(), lr=0.02, momentum=0.9)
x = torch.from_numpy(synth.gauss_ball(32, 1024, 1)).cuda(); f = torch.from_numpy(synth.unit_normals(32, 1024, 2)).cuda()
y = torch.from_numpy(synth.labels(32, 40, 3)).cuda()
params = list(net.parameters())
def step():
    for p in params: p.grad = None
    soft_cross_entropy_loss(net(x, f), y).backward(); opt.step()
for _ in range(10): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(40):
    step()
    if i % 4 == 3: torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 30)
