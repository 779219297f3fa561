"""Does the occasional 8 % slower mode of a process follow the memory placement?  Re-allocate everything a few times
inside one process and time each placement."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from pointcloudlib_amd import synth
from pointcloudlib_amd.networks.cls.pointnet2 import PointNet2_cls
from pointcloudlib_amd.train_utils import soft_cross_entropy_loss, make_sgd
torch.manual_seed(0)
net = PointNet2_cls().cuda().train()
opt = make_sgd(net.parameters(), lr=0.02, momentum=0.9)
B, N = 32, 1024
batches = [(torch.from_numpy(synth.gauss_ball(B, N, 1 + i)).cuda(), torch.from_numpy(synth.unit_normals(B, N, 9 + i)).cuda(),
            torch.from_numpy(synth.labels(B, 40, 3 + i)).cuda()) for i in range(4)]
side = torch.cuda.Stream(priority=-1)
pending = {}
def step(i):
    x, f, y = batches[i % 4]
    for p in net.parameters(): p.grad = None
    out = net(x, f, sampling=pending.pop(i, None))
    pending[i + 1] = net.precompute_sampling(batches[(i + 1) % 4][0], stream=side)
    soft_cross_entropy_loss(out, y).backward(); opt.step()
it = 0
for rep in range(8):
    torch.cuda.synchronize(); pending.clear(); torch.cuda.empty_cache()
    for _ in range(20): step(it); it += 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(60): step(it); it += 1
    torch.cuda.synchronize(); print(f"placement {rep}: {(time.perf_counter() - t0) / 60 * 1e3:.3f} ms/step", flush=True)
