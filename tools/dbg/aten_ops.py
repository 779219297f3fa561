"""Which aten ops (PyTorch's own kernels) does one training step of a network still launch, and from which line of the package?
    python tools/dbg/aten_ops.py [pointnet2 | msg | dgcnn | pointconv | pointcnn]
Every such op is a dependent launch (~5-9 us) the library could absorb; used to find the glue removed in round 3."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from pointcloudlib_amd import synth  # noqa: E402
from pointcloudlib_amd.train_utils import make_sgd, seg_cross_entropy_loss, soft_cross_entropy_loss  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "pointnet2"
torch.manual_seed(0)
B, N = (16, 2048) if which == "msg" else (32, 1024)
x = torch.from_numpy(synth.gauss_ball(B, N, 1)).cuda()
f = torch.from_numpy(synth.unit_normals(B, N, 2)).cuda()
y = torch.from_numpy(synth.labels(B, 40, 3)).cuda()
if which == "pointnet2":
    from pointcloudlib_amd.networks.cls.pointnet2 import PointNet2_cls
    net = PointNet2_cls().cuda().train()
    fwd = lambda: soft_cross_entropy_loss(net(x, f), y)
elif which == "msg":
    from pointcloudlib_amd.networks.seg.pointnet2_partseg import PointNetMSG
    net = PointNetMSG().cuda().train()
    onehot = torch.zeros(B, 16, device="cuda"); onehot[:, 3] = 1
    seg = torch.randint(0, 50, (B, N), device="cuda")
    fwd = lambda: seg_cross_entropy_loss(net(x, f, onehot), seg)
elif which == "dgcnn":
    from pointcloudlib_amd.networks.cls.dgcnn import DGCNN
    net = DGCNN().cuda().train()
    xt = x.transpose(1, 2).contiguous()
    fwd = lambda: soft_cross_entropy_loss(net(xt), y)
elif which == "pointconv":
    from pointcloudlib_amd.networks.cls.pointconv import PointConvDensityClsSsg
    net = PointConvDensityClsSsg().cuda().train()
    xt = x.transpose(1, 2).contiguous()
    fwd = lambda: soft_cross_entropy_loss(net(xt), y)
elif which == "pointcnn":
    from pointcloudlib_amd.networks.cls.pointcnn import PointCNNcls
    net = PointCNNcls().cuda().train()
    fwd = lambda: soft_cross_entropy_loss(net(x), y)
else:
    raise SystemExit(f"unknown network {which!r}")
opt = make_sgd(net.parameters(), lr=0.02, momentum=0.9)


def step():
    for p in net.parameters():
        p.grad = None
    fwd().backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step()
    torch.cuda.synchronize()
trivial = {"aten::detach", "aten::view", "aten::reshape", "aten::unsqueeze", "aten::empty", "aten::empty_like", "aten::split_with_sizes",
           "aten::alias", "aten::transpose", "aten::permute", "aten::expand", "aten::as_strided", "aten::t", "aten::squeeze", "aten::select",
           "aten::slice", "aten::_unsafe_view", "aten::empty_strided", "aten::result_type", "aten::view_as", "aten::unbind", "aten::item",
           "aten::_local_scalar_dense", "aten::is_nonzero", "aten::narrow", "aten::unflatten"}
cnt = collections.Counter()
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::") or e.name in trivial:
        continue
    if e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
        continue
    par = e.cpu_parent.name if e.cpu_parent is not None else "-"
    where = [s for s in (e.stack or []) if "pointcloudlib_amd" in s or "train_utils" in s][:1]
    cnt[(e.name, par[:40], where[0].split("pointcloudlib_amd/")[-1][:60] if where else "")] += 1
print(f"{which}: ops with a kernel behind them, per step (op, autograd parent, nearest package frame)")
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(f"  {v:3d}  {k}")
