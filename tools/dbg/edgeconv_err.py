"""One EdgeConv stage of DGCNN in isolation (stage 4: 128 -> 256, k = 20, B = 32, N = 1024): the HIP path's input gradient and weight
gradient against the fp64 edge-tensor formulation, with the stage's input and upstream gradient taken from the fp64 restatement
-- beside the same stage in PyTorch-CPU fp32 (edge tensor) on the same inputs."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
import oracle; oracle.build()
import oracle.torch_backend  # noqa
from oracle.cpu_dgcnn import DGCNNCPU
from pointcloudlib_amd import synth
from pointcloudlib_amd.misc.edgeconv import edge_conv
from pointcloudlib_amd.networks.cls.dgcnn import DGCNN
from pointcloudlib_amd.train_utils import soft_cross_entropy_loss
B, N, k = int(os.environ.get("B", 32)), 1024, 20
torch.manual_seed(0)
dev = torch.device("cuda")
net = DGCNN().to(dev).train()
state = net.state_dict()
r64 = DGCNNCPU(state, k, dtype=torch.float64)
x = torch.from_numpy(synth.gauss_ball(B, N, 20243)).transpose(1, 2).contiguous()
lab = torch.from_numpy(synth.labels(B, 40, 21143))
l64, a64 = r64(x, return_aux=True)
for f in a64["feats"]: f.retain_grad()
soft_cross_entropy_loss(l64, lab).backward()
rel = lambda a, b: ((a.double().cpu() - b).norm() / b.norm()).item()
for S in (1, 2, 3, 4):
    name = f"conv{S}"
    xin64 = (x.transpose(1, 2).double() if S == 1 else a64["feats"][S - 2].detach())
    g64 = a64["feats"][S - 1].grad.detach()
    idx = a64["lists"][S - 1]
    W, gam, bet = (state[f"{name}.{n}.0"].double().cpu() for n in ("weights", "gammas", "betas"))
    C = xin64.shape[2]; Co = W.shape[0]; bi = torch.arange(B)[:, None, None]
    def edge(dt):
        xi = xin64.detach().clone().to(dt).requires_grad_(True); w = W.detach().clone().to(dt).requires_grad_(True)
        nb = xi[bi, idx]; ctr = xi[:, :, None, :].expand(B, N, k, C); e = torch.cat([nb - ctr, ctr], -1)
        y = torch.nn.functional.linear(e.reshape(-1, 2 * C), w)
        z = torch.nn.functional.leaky_relu(torch.nn.functional.batch_norm(y, None, None, gam.to(dt), bet.to(dt), True, 0.0, 1e-5), 0.2)
        out = z.reshape(B, N, k, Co).max(2)[0]
        out.backward(g64.to(dt))
        return out.detach(), xi.grad, w.grad
    o64, dx64, dw64 = edge(torch.float64)
    o32, dx32, dw32 = edge(torch.float32)
    mlp = getattr(net, name)
    mlp.zero_grad()
    xh = xin64.float().to(dev).requires_grad_(True)
    oh = edge_conv(mlp, xh, idx.to(dev).int().contiguous())
    oh.backward(g64.float().to(dev))
    print(f"stage {S} (C={C}->{Co}): out hip {rel(oh.detach(), o64):.2e} / cpu-fp32 {rel(o32, o64):.2e} | dx hip {rel(xh.grad, dx64):.2e} / cpu-fp32 {rel(dx32, dx64):.2e} | "
          f"dW hip {rel(mlp.weights[0].grad, dw64):.2e} / cpu-fp32 {rel(dw32, dw64):.2e}", flush=True)
    # how many max winners differ from the fp64 evaluation
