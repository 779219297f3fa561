"""DGCNN whole-network gradients against fp64 for several weight / input seeds: on how many of the 23 parameter tensors is the HIP network
closer to fp64 than the PyTorch-CPU fp32 restatement, and how many max-pool winners of stage 4 differ from the fp64 evaluation's?
(The count is decided by ~10 winner flips out of 8.4 M per stage: it moves with the seed, not with the arithmetic.)"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import oracle; oracle.build()
from oracle.cpu_dgcnn import DGCNNCPU
from pointcloudlib_amd import synth
from pointcloudlib_amd.networks.cls.dgcnn import DGCNN
from pointcloudlib_amd.train_utils import soft_cross_entropy_loss
B, N, k = 32, 1024, 20
dev = torch.device("cuda")
rel = lambda a, b: ((a.double().cpu() - b).norm() / b.norm()).item()
for seed in [int(s) for s in (sys.argv[1:] or ["0", "1", "2", "3"])]:
    torch.manual_seed(seed)
    pts, lab = synth.gauss_ball(B, N, 20243 + seed), synth.labels(B, 40, 21143 + seed)
    net = DGCNN().to(dev).train()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout): m.p = 0.0
    state = net.state_dict()
    r32, r64 = DGCNNCPU(state, k), DGCNNCPU(state, k, dtype=torch.float64)
    xin = torch.from_numpy(pts).transpose(1, 2).contiguous()
    l32, a32 = r32(xin, return_aux=True)
    lists = a32["lists"]
    l64, a64 = r64(xin, lists=lists, return_aux=True)
    y = torch.from_numpy(lab)
    soft_cross_entropy_loss(l32, y).backward(); soft_cross_entropy_loss(l64, y).backward()
    out = net(xin.to(dev), lists=[l.to(dev).int().contiguous() for l in lists])
    soft_cross_entropy_loss(out, y.to(dev)).backward()
    closer, worst_h, worst_c, n = 0, 0.0, 0.0, 0
    for name, p in net.named_parameters():
        g64 = r64.grad(name)
        if g64.abs().max() < 1e-12: continue
        eh, ec = rel(p.grad, g64), rel(r32.grad(name), g64)
        closer += eh < ec; n += 1; worst_h = max(worst_h, eh); worst_c = max(worst_c, ec)
    print(f"seed {seed}: HIP closer to fp64 than PyTorch-CPU fp32 on {closer} of {n} gradient tensors; worst relL2 hip {worst_h:.2e} / fp32 restatement {worst_c:.2e}; "
          f"logits max|err| hip {(out.detach().cpu().double() - l64.detach()).abs().max().item():.2e} / fp32 restatement {(l32.detach().double() - l64.detach()).abs().max().item():.2e}", flush=True)
