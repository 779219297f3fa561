"""DGCNN stage 4 in the whole network: is the gradient's distance from fp64 decided by the stage's INPUT (x3 of each pipeline) or by the
stage's own arithmetic?  The fp64 edge-form stage is evaluated on (a) the fp64 network's x3, (b) the HIP network's x3, (c) the PyTorch-CPU
fp32 network's x3 -- same lists, same upstream gradient -- and the input gradients compared with (a).  Beside them the actual dL/dx3 of the
HIP network and of the fp32 restatement."""
import os, sys, torch
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
xin = torch.from_numpy(pts).transpose(1, 2).contiguous()
l32, a32 = r32(xin, return_aux=True)
lists = a32["lists"]
l64, a64 = r64(xin, lists=lists, return_aux=True)
for f in a32["feats"] + a64["feats"]: f.retain_grad()
y = torch.from_numpy(lab)
soft_cross_entropy_loss(l32, y).backward(); soft_cross_entropy_loss(l64, y).backward()
out, stages = net(xin.to(dev), lists=[l.to(dev).int().contiguous() for l in lists], return_stages=True)
for s in stages: s.retain_grad()
soft_cross_entropy_loss(out, y.to(dev)).backward()
rel = lambda a, b: ((a.double().cpu() - b).norm() / b.norm()).item()
for S in (2, 3, 4):
    idx = lists[S - 1].long(); g64 = a64["feats"][S - 1].grad.detach()
    W, gam, bet = (state[f"conv{S}.{n}.0"].double().cpu() for n in ("weights", "gammas", "betas"))
    bi = torch.arange(B)[:, None, None]
    def stage64(x3):
        xi = x3.detach().double().cpu().clone().requires_grad_(True)
        C = xi.shape[2]
        nb = xi[bi, idx]; ctr = xi[:, :, None, :].expand(B, N, k, C); e = torch.cat([nb - ctr, ctr], -1)
        yv = torch.nn.functional.linear(e.reshape(-1, 2 * C), W)
        z = torch.nn.functional.leaky_relu(torch.nn.functional.batch_norm(yv, None, None, gam, bet, True, 0.0, 1e-5), 0.2).reshape(B, N, k, -1)
        o, arg = z.max(2)
        o.backward(g64)
        return xi.grad, arg
    da, aa = stage64(a64["feats"][S - 2]); dh, ah = stage64(stages[S - 2]); dc, ac = stage64(a32["feats"][S - 2])
    print(f"stage {S}: fp64 stage on the HIP net's input: dx {rel(dh, da):.2e} ({(ah != aa).float().mean().item():.2e} of the winners differ) | on the fp32 restatement's input: dx {rel(dc, da):.2e} "
          f"({(ac != aa).float().mean().item():.2e}) | actual dL/dx{S - 1}: hip {rel(stages[S - 2].grad, a64['feats'][S - 2].grad):.2e}, fp32 restatement {rel(a32['feats'][S - 2].grad, a64['feats'][S - 2].grad):.2e}"
          f" | input distance from fp64: hip {rel(stages[S - 2], a64['feats'][S - 2].detach()):.2e}, fp32 restatement {rel(a32['feats'][S - 2], a64['feats'][S - 2].detach()):.2e}", flush=True)
