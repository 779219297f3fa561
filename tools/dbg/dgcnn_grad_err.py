"""Where does DGCNN's gradient error enter?  Relative L2 error against the fp64 restatement of (a) the gradient w.r.t. every stage
output x1..x4 and (b) every parameter gradient, for the HIP network and for the PyTorch-CPU fp32 restatement, on shared lists."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
import oracle; oracle.build()
from oracle.cpu_dgcnn import DGCNNCPU
from pointcloudlib_amd import synth
from pointcloudlib_amd.networks.cls.dgcnn import DGCNN
from pointcloudlib_amd.train_utils import soft_cross_entropy_loss
B, N, k = 32, 1024, 20
torch.manual_seed(0)
dev = torch.device("cuda")
pts, lab = synth.gauss_ball(B, N, 20243), synth.labels(B, 40, 21143)
net = DGCNN().to(dev).train()
for m in net.modules():
    if isinstance(m, torch.nn.Dropout): m.p = 0.0
state = net.state_dict()
r32, r64 = DGCNNCPU(state, k), DGCNNCPU(state, k, dtype=torch.float64)
xin_cpu = torch.from_numpy(pts).transpose(1, 2).contiguous()
l32, a32 = r32(xin_cpu, return_aux=True)
lists = a32["lists"]
l64, a64 = r64(xin_cpu, lists=lists, return_aux=True)
for f in a32["feats"] + a64["feats"]: f.retain_grad()
y = torch.from_numpy(lab)
soft_cross_entropy_loss(l32, y).backward(); soft_cross_entropy_loss(l64, y).backward()
out, stages = net(xin_cpu.to(dev), lists=[l.to(dev).int().contiguous() for l in lists], return_stages=True)
for s in stages: s.retain_grad()
soft_cross_entropy_loss(out, y.to(dev)).backward()
rel = lambda a, b: ((a.double().cpu() - b).norm() / b.norm()).item()
for i in range(4):
    print(f"dL/dx{i + 1}: hip {rel(stages[i].grad, a64['feats'][i].grad):.2e}  r32 {rel(a32['feats'][i].grad, a64['feats'][i].grad):.2e}")
for n, p in net.named_parameters():
    g64 = r64.grad(n)
    if g64.abs().max() < 1e-12: continue
    print(f"{n:24s} hip {rel(p.grad, g64):.2e}  r32 {rel(r32.grad(n), g64):.2e}")
