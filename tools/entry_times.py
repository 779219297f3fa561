"""Per C-ABI entry point and shape: time of one train step of a model (KernelTimer, HIP events).
    python tools/entry_times.py pointconv|dgcnn|pointnet2_seg|... [rows]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudlib_amd import _lib, synth
from pointcloudlib_amd.train_utils import make_sgd, soft_cross_entropy_loss
which = sys.argv[1]; rows = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = "cuda"; torch.manual_seed(0)
B, N = 32, 1024
x = torch.from_numpy(synth.gauss_ball(B, N, 1)).to(dev); xt = x.transpose(1, 2).contiguous()
y = torch.from_numpy(synth.labels(B, 40, 2)).to(dev)
if which == "pointconv":
    from pointcloudlib_amd.networks.cls.pointconv import PointConvDensityClsSsg as M; args = (xt,)
elif which == "dgcnn":
    from pointcloudlib_amd.networks.cls.dgcnn import DGCNN as M; args = (xt,)
elif which == "pointcnn":
    from pointcloudlib_amd.networks.cls.pointcnn import PointCNNcls as M; args = (x,)
net = M().to(dev).train(); opt = make_sgd(net.parameters(), lr=0.02, momentum=0.9)
def step():
    opt.zero_grad(set_to_none=True); soft_cross_entropy_loss(net(*args), y).backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
t = _lib.KernelTimer(); _lib.PROFILER = t
K = 5
for _ in range(K): step()
torch.cuda.synchronize(); _lib.PROFILER = None
s = t.summary()
tot = sum(v["total_ms"] for v in s.values()) / K
print(f"{which}: own C-ABI calls {tot:.3f} ms/step")
for (name, tag), v in sorted(s.items(), key=lambda kv: -kv[1]["total_ms"])[:rows]:
    print(f"{v['total_ms'] / K * 1e3:8.1f} us  x{v['launches'] / K:4.1f}  avg {v['avg_ms'] * 1e3:7.1f}  {name:30s} {tag}")
