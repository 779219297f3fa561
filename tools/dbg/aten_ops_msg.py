"""Which aten ops (torch's own kernels) does one PointNet++ MSG part-seg step launch, and from where?"""
import os, sys, torch, collections
sys.path.insert(0, os.getcwd())
from pointcloudlib_amd import synth
from pointcloudlib_amd.networks.seg.pointnet2_partseg import PointNetMSG
from pointcloudlib_amd.train_utils import make_sgd
torch.manual_seed(0)
net = PointNetMSG().cuda().train()
opt = make_sgd(net.parameters(), lr=0.02, momentum=0.9)
B, N = 16, 2048
x = torch.from_numpy(synth.gauss_ball(B, N, 1)).cuda(); f = torch.from_numpy(synth.unit_normals(B, N, 2)).cuda()
onehot = torch.zeros(B, 16, device="cuda"); onehot[:, 3] = 1
y = torch.randint(0, 50, (B, N), device="cuda")
def step():
    for p in net.parameters(): p.grad = None
    torch.nn.functional.cross_entropy(net(x, f, onehot), y).backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step(); torch.cuda.synchronize()
trivial = {"aten::detach", "aten::view", "aten::reshape", "aten::unsqueeze", "aten::empty", "aten::empty_like", "aten::split_with_sizes", "aten::alias",
           "aten::transpose", "aten::permute", "aten::expand", "aten::as_strided", "aten::t", "aten::squeeze", "aten::select", "aten::slice", "aten::_unsafe_view",
           "aten::empty_strided", "aten::result_type", "aten::view_as", "aten::unbind", "aten::item", "aten::_local_scalar_dense", "aten::is_nonzero", "aten::narrow", "aten::unflatten"}
cnt = collections.Counter()
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::") or e.name in trivial:
        continue
    if e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
        continue
    par = e.cpu_parent.name if e.cpu_parent is not None else "-"
    where = [s for s in (e.stack or []) if "pointcloudlib_amd" in s][:1]
    cnt[(e.name, par[:40], where[0].split("pointcloudlib_amd/")[-1][:60] if where else "")] += 1
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(v, k)
