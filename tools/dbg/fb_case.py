import copy, os, sys, torch
sys.path.insert(0, ".")
from pointcloudlib_amd.misc.layers import PointwiseMLP
from pointcloudlib_amd.misc import mlp_hip
from tests.test_mlp_hip import run

def case(spec, lead, ns, bias, slope, fused):
    mlp_hip._FUSED_BWD = fused
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
    err = lambda a, b: (a.double().cpu() - b).abs().max().item()
    msg = [f"x {err(h[1], ref[1]):.2e}/{ref[1].abs().max():.2e}"]
    for n in ref[2]:
        if "weights" in n or "gammas" in n:
            msg.append(f"{n} {err(h[2][n], ref[2][n]):.2e}/{ref[2][n].abs().max():.2e}")
    print(spec, lead, ns, slope, "fused" if fused else "split", " ".join(msg), flush=True)

for fused in (False, True):
    case([16, 64, 128, 64], (3, 11111), None, False, 0.2, fused)
    case([16, 64, 128, 64], (3, 11111), None, False, 0.0, fused)
    case([16, 64, 128, 64], (3, 11136), None, False, 0.2, fused)
    case([16, 64, 128, 64], (1, 16384), None, False, 0.0, fused)
    case([16, 64, 128, 64], (1, 16320), None, False, 0.0, fused)
    case([16, 64, 128], (1, 33333), None, False, 0.0, fused)
    case([16, 128, 64], (1, 33333), None, False, 0.0, fused)
