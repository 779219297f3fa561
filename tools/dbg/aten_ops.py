"""Which aten ops (torch's own kernels) does one PointNet++ step still launch?  torch.profiler, CPU-op -> kernel list."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from pointcloudlib_amd import synth
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
for _ in range(5): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::")]
import collections
cnt = collections.Counter(e.name for e in evs if not (e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::")))
print(cnt)
trivial = {"aten::detach", "aten::view", "aten::reshape", "aten::unsqueeze", "aten::empty", "aten::empty_like", "aten::split_with_sizes", "aten::alias",
           "aten::transpose", "aten::permute", "aten::expand", "aten::as_strided", "aten::t"}
for e in evs:
    if e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
        continue
    if e.name in trivial:
        continue
    par = e.cpu_parent.name if e.cpu_parent is not None else None
    print(f"{e.name:28s} parent={par}  t={e.time_range.start}")
