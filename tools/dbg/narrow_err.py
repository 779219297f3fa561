"""Error of the narrow-stack kernels / the GEMM path / PyTorch fp32 against fp64, tensor by tensor (large-mean input)."""
import copy, os, sys, torch
sys.path.insert(0, os.getcwd())
import oracle.torch_backend  # noqa: F401  (registers the plain-PyTorch composite)
from pointcloudlib_amd.misc.layers import PointwiseMLP
from pointcloudlib_amd.misc import mlp_hip
spec, rows = [3, 8, 8, 16], int(sys.argv[1]) if len(sys.argv) > 1 else 5000
offset, spread = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.3, 1e-3)
def run(m, x, g, backend):
    m.backend = backend; m.zero_grad(); out = m(x); out.backward(g)
    return out.detach(), {n: p.grad.detach().clone() for n, p in m.named_parameters()}
torch.manual_seed(7 + rows + spec[0])
m64 = PointwiseMLP(spec, bias=True, slope=0.0).double()
with torch.no_grad():
    for g, b in zip(m64.gammas, m64.betas):
        g.uniform_(0.5, 1.5); b.uniform_(-0.3, 0.3)
    m64.gammas[1][::3] *= -1.0
x64 = offset + spread * torch.randn(rows, spec[0], dtype=torch.float64)
g64 = torch.randn(rows, spec[-1], dtype=torch.float64)
ref = run(copy.deepcopy(m64), x64, g64, "torch")
m32 = copy.deepcopy(m64).float().cuda().train()
x32, g32 = x64.float().cuda(), g64.float().cuda()
t = run(copy.deepcopy(m32), x32, g32, "torch")
h = run(copy.deepcopy(m32), x32, g32, "hip")
with mlp_hip.per_kernel_path():
    o = run(copy.deepcopy(m32), x32, g32, "hip")
e = lambda a, b: (a.double().cpu() - b).abs().max().item()
print(f"out: narrow {e(h[0], ref[0]):.2e} gemm {e(o[0], ref[0]):.2e} torch {e(t[0], ref[0]):.2e}  flips narrow {((h[0].cpu() > 0) != (ref[0] > 0)).sum().item()} gemm {((o[0].cpu() > 0) != (ref[0] > 0)).sum().item()} torch {((t[0].cpu() > 0) != (ref[0] > 0)).sum().item()}")
for n in ref[1]:
    s = ref[1][n].abs().max().item()
    print(f"{n:12s} scale {s:9.3e}  narrow {e(h[1][n], ref[1][n]):.2e} gemm {e(o[1][n], ref[1][n]):.2e} torch {e(t[1][n], ref[1][n]):.2e}")
