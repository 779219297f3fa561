"""Per-entry-point event timing of one train step of a model (dgcnn | partseg | pointconv | pointnet).
    python tools/profile_model.py dgcnn"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudlib_amd import _lib, synth
from pointcloudlib_amd.train_utils import soft_cross_entropy_loss
which = sys.argv[1] if len(sys.argv) > 1 else "dgcnn"
dev = "cuda"
torch.manual_seed(0)
if which == "dgcnn":
    from pointcloudlib_amd.networks.cls.dgcnn import DGCNN
    net = DGCNN().to(dev).train(); B = 32
    x = torch.from_numpy(synth.gauss_ball(B, 1024, 1)).to(dev).transpose(1, 2).contiguous(); args = (x,)
    y = torch.from_numpy(synth.labels(B, 40, 2)).to(dev); loss_fn = lambda o: soft_cross_entropy_loss(o, y)
elif which == "pointconv":
    from pointcloudlib_amd.networks.cls.pointconv import PointConvDensityClsSsg
    net = PointConvDensityClsSsg().to(dev).train(); B = 32
    x = torch.from_numpy(synth.gauss_ball(B, 1024, 1)).to(dev).transpose(1, 2).contiguous(); args = (x,)
    y = torch.from_numpy(synth.labels(B, 40, 2)).to(dev); loss_fn = lambda o: soft_cross_entropy_loss(o, y)
elif which == "partseg":
    from pointcloudlib_amd.networks.seg.pointnet2_partseg import PointNet2_partseg
    net = PointNet2_partseg().to(dev).train(); B = 16
    x = torch.from_numpy(synth.gauss_ball(B, 2048, 1)).to(dev)
    oh = torch.zeros(B, 16, device=dev); oh[torch.arange(B), torch.arange(B) % 16] = 1; args = (x, x, oh)
    seg = torch.randint(0, 50, (B, 2048), device=dev); loss_fn = lambda o: torch.nn.functional.cross_entropy(o, seg)
opt = torch.optim.SGD(net.parameters(), lr=0.02, momentum=0.9)
def step():
    opt.zero_grad(set_to_none=True); loss_fn(net(*args)).backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print(f"{which}: {(time.perf_counter()-t0)*100:.3f} ms/step")
_lib.PROFILER = _lib.KernelTimer()
for _ in range(3): step()
torch.cuda.synchronize()
summ = _lib.PROFILER.summary(); _lib.PROFILER = None
tot = sum(v["total_ms"] for v in summ.values()) / 3
print(f"own C-ABI calls per step: {tot:.3f} ms")
for (name, tag), v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])[:28]:
    gbs = v["algo_bytes"] / (v["avg_ms"] * 1e-3) / 1e9 if v["algo_bytes"] else 0
    tf = v["algo_flops"] / (v["avg_ms"] * 1e-3) / 1e12 if v["algo_flops"] else 0
    print(f"{name:28s} {tag:16s} n/step={v['launches']/3:4.1f} avg={v['avg_ms']:8.4f} ms {gbs:8.1f} GB/s {tf:7.2f} TF")
