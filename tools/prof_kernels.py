"""Top GPU kernels of one train step of a model (torch.profiler, device time).
    python tools/prof_kernels.py pointcnn|pointcnn_seg|pointconv|pointconv_seg|dgcnn|dgcnn_seg|pointnet_seg [rows]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudlib_amd import synth
from pointcloudlib_amd.train_utils import make_sgd, soft_cross_entropy_loss
from pointcloudlib_amd.affinity import pin_to_gpu_node
pin_to_gpu_node(0)
which = sys.argv[1]
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 25
dev = "cuda"
torch.manual_seed(0)
ce = torch.nn.functional.cross_entropy
def cloud(B, N): return torch.from_numpy(synth.gauss_ball(B, N, 1)).to(dev)
if which.endswith("_seg"):
    B, N = 16, 2048
    x = cloud(B, N); xt = x.transpose(1, 2).contiguous()
    oh = torch.zeros(B, 16, device=dev); oh[torch.arange(B), torch.arange(B) % 16] = 1
    seg = torch.randint(0, 50, (B, N), device=dev)
    if which == "pointcnn_seg":
        from pointcloudlib_amd.networks.seg.pointcnn_partseg import PointCNN_partseg as M; args = (x,); lf = lambda o: ce(o, seg)
    elif which == "pointconv_seg":
        from pointcloudlib_amd.networks.seg.pointconv_partseg import PointConvDensity_partseg as M; args = (x, oh); lf = lambda o: ce(o.permute(0, 2, 1), seg)
    elif which == "dgcnn_seg":
        from pointcloudlib_amd.networks.seg.dgcnn_partseg import DGCNN_partseg; M = lambda: DGCNN_partseg(50); args = (xt, oh); lf = lambda o: ce(o, seg)
    elif which == "pointnet2_seg":
        from pointcloudlib_amd.networks.seg.pointnet2_partseg import PointNet2_partseg as M; args = (x, x, oh); lf = lambda o: ce(o, seg)
    elif which == "pointnet2msg_seg":
        from pointcloudlib_amd.networks.seg.pointnet2_partseg import PointNetMSG as M; args = (x, x, oh); lf = lambda o: ce(o, seg)
    elif which == "pointnet_seg":
        from pointcloudlib_amd.networks.seg.pointnet_partseg import PointNet_partseg as M; args = (xt, oh); lf = lambda o: ce(o, seg)
else:
    B, N = 32, 1024
    x = cloud(B, N); xt = x.transpose(1, 2).contiguous()
    y = torch.from_numpy(synth.labels(B, 40, 2)).to(dev); lf = lambda o: soft_cross_entropy_loss(o, y)
    if which == "pointcnn":
        from pointcloudlib_amd.networks.cls.pointcnn import PointCNNcls as M; args = (x,)
    elif which == "pointconv":
        from pointcloudlib_amd.networks.cls.pointconv import PointConvDensityClsSsg as M; args = (xt,)
    elif which == "dgcnn":
        from pointcloudlib_amd.networks.cls.dgcnn import DGCNN as M; args = (xt,)
net = M().to(dev).train()
opt = make_sgd(net.parameters(), lr=0.02, momentum=0.9)
def step():
    opt.zero_grad(set_to_none=True); lf(net(*args)).backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print(f"{which}: {(time.perf_counter() - t0) * 100:.3f} ms/step")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
ev = [e for e in prof.key_averages() if e.device_type.name == "CUDA" or (getattr(e, "self_device_time_total", 0) > 0 and e.key.startswith(("void", "pcl::", "Cijk", "__amd")))]
ev = [e for e in prof.key_averages() if e.self_device_time_total > 0 and not e.key.startswith(("aten::", "autograd::", "Optimizer", "_", "torch::", "detach"))]
tot = sum(e.self_device_time_total for e in ev) / 3
print(f"device kernel time per step: {tot / 1e3:.3f} ms")
for e in sorted(ev, key=lambda e: -e.self_device_time_total)[:rows]:
    print(f"{e.self_device_time_total / 3:9.1f} us  x{e.count / 3:5.1f}  {e.key[:110]}")
