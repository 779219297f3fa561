import os, sys, time, torch, cProfile, pstats
sys.path.insert(0, os.getcwd())
from pointcloudlib_amd.affinity import pin_to_gpu_node; pin_to_gpu_node(0)
from pointcloudlib_amd import synth
from pointcloudlib_amd.networks.seg.pointnet2_partseg import PointNet2_partseg
from pointcloudlib_amd.train_utils import make_sgd
torch.manual_seed(0)
B, N = 16, 2048
net = PointNet2_partseg().cuda().train()
opt = make_sgd(net.parameters(), lr=0.02, momentum=0.9)
x = torch.from_numpy(synth.gauss_ball(B, N, 1)).cuda()
oh = torch.zeros(B, 16, device="cuda"); oh[torch.arange(B), torch.arange(B) % 16] = 1
seg = torch.randint(0, 50, (B, N), device="cuda")
ce = torch.nn.functional.cross_entropy
def step():
    opt.zero_grad(set_to_none=True); ce(net(x, x, oh), seg).backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    for _ in range(20): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0)/20:.3f} ms/step   total {1e3*(t2-t0)/20:.3f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(40)
