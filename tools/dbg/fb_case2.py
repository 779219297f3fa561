import copy, os, sys, torch
sys.path.insert(0, ".")
from pointcloudlib_amd.misc.layers import PointwiseMLP
from tests.test_mlp_hip import run

def case(spec, lead, ns, bias, slope):
    torch.manual_seed(1234 + spec[0])
    m64 = PointwiseMLP(spec, bias=bias, slope=slope).double()
    with torch.no_grad():
        for g, b in zip(m64.gammas, m64.betas):
            g.uniform_(0.5, 1.5); b.uniform_(-0.3, 0.3)
    x64 = torch.randn(*lead, spec[0], dtype=torch.float64)
    m64.backend = "torch"
    out_shape = (*lead[:-1], spec[-1]) if ns else (*lead, spec[-1])
    gout64 = torch.randn(out_shape, dtype=torch.float64)
    ref = run(copy.deepcopy(m64), x64, ns, gout64, "torch")
    m32 = copy.deepcopy(m64).float().cuda()
    h = run(copy.deepcopy(m32), x64.float().cuda(), ns, gout64.float().cuda(), "hip")
    t = run(copy.deepcopy(m32), x64.float().cuda(), ns, gout64.float().cuda(), "torch")
    for name, r in (("hip", h), ("torch32", t)):
        e = (r[1].double().cpu() - ref[1]).abs().reshape(-1, spec[0]).max(1).values
        eo = (r[0].double().cpu() - ref[0]).abs().max().item()
        bad = (e > 1e-4).nonzero().flatten()
        print(name, spec, lead, "feat err", f"{eo:.2e}", "x-grad rows>1e-4:", bad.numel(), bad[:8].tolist(), "median row err", f"{e.median():.2e}", "max", f"{e.max():.2e}", flush=True)
    # margin of the fp64 reference's pre-activations
    with torch.no_grad():
        z = x64.reshape(-1, spec[0])
        for l in range(len(spec) - 1):
            y = z @ m64.weights[l].t()
            mu, var = y.mean(0), y.var(0, unbiased=False)
            zz = (y - mu) / torch.sqrt(var + 1e-5) * m64.gammas[l] + m64.betas[l]
            k = zz.abs().argmin()
            print("  layer", l, "min |pre-activation|", f"{zz.abs().min():.2e}", "at row", (k // zz.shape[1]).item())
            z = torch.nn.functional.leaky_relu(zz, slope)

case([16, 128, 64], (1, 33333), None, False, 0.0)
case([16, 64, 128, 64], (3, 11136), None, False, 0.2)
