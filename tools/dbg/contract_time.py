"""Time PointConv's contraction entry points alone at the three levels' sizes (events, 20 reps)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from pointcloudlib_amd import _lib
from pointcloudlib_amd.misc.ops import _p, _stream
dev = torch.device("cuda")
def t(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for name, G, ns, C in (("sa1", 16384, 32, 64), ("sa2", 4096, 64, 128), ("sa3", 32, 128, 256)):
    NSET = 6 if name != "sa3" else 1           # rotate over 6 sets of buffers (1.4 GB): every call reads cold HBM, as inside a training step
    Ys = [torch.randn(G * ns, C, device=dev) for _ in range(NSET)]; outs = [torch.empty(G, C * 16, device=dev) for _ in range(NSET)]
    ws = [torch.randn(G * ns, 16, device=dev) for _ in range(NSET)]
    Y = Ys[0]; sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev) * 0.1
    dens = torch.rand(G * ns, device=dev) + 0.5; w = torch.randn(G * ns, 16, device=dev)
    out = torch.empty(G, C * 16, device=dev); dout = torch.randn(G, C * 16, device=dev)
    du = torch.empty_like(Y); dw = torch.empty_like(w); dd = torch.empty_like(dens)
    rows = _lib.lib().pcl_pointconv_contract_bn_stat_rows(G)
    st = torch.empty(rows * 2 * C, dtype=torch.float64, device=dev)
    f = lambda: _lib.call("pcl_pointconv_contract_bn_f32", _p(Y), _p(sc), _p(sh), 0.0, _p(dens), _p(w), G, ns, C, 16, _p(out), _stream())
    b = lambda: _lib.call("pcl_pointconv_contract_bn_bwd_f32", _p(dout), _p(Y), _p(sc), _p(sh), 0.0, _p(dens), _p(w), G, ns, C, 16, _p(du), _p(dw), _p(dd), _p(st), _stream())
    mb = (Y.numel() + w.numel() + out.numel()) * 4 / 1e6
    cnt = [0]
    def fc():
        i = cnt[0] % NSET; cnt[0] += 1
        _lib.call("pcl_pointconv_contract_bn_f32", _p(Ys[i]), _p(sc), _p(sh), 0.0, _p(dens), _p(ws[i]), G, ns, C, 16, _p(outs[i]), _stream())
    tf, tb, tfc = t(f), t(b), t(fc, 24)
    print(f"   cold rotation: fwd {tfc:7.1f} us")
    print(f"{name}: G={G} ns={ns} C={C}  fwd {tf:7.1f} us ({mb / tf * 1e-6 * 1e6 / 1e6:.2f} TB/s of {mb:.0f} MB)   bwd (feat + w) {tb:7.1f} us")

# the same forward call right after a heavy fp32 GEMM (as inside a training step): is the slowdown seen in the kernel traces a
# clock / power effect?
A = torch.randn(8192, 8192, device=dev); Bm = torch.randn(8192, 8192, device=dev)
G, ns, C = 16384, 32, 64
Y = torch.randn(G * ns, C, device=dev); sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev) * 0.1
dens = torch.rand(G * ns, device=dev) + 0.5; w = torch.randn(G * ns, 16, device=dev); out = torch.empty(G, C * 16, device=dev)
for heavy in (0, 1, 4):
    ts = []
    for rep in range(12):
        for _ in range(heavy): torch.matmul(A, Bm)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.call("pcl_pointconv_contract_bn_f32", _p(Y), _p(sc), _p(sh), 0.0, _p(dens), _p(w), G, ns, C, 16, _p(out), _stream())
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    print(f"sa1 fwd after {heavy} x 8192^3 fp32 matmul: median {ts[len(ts) // 2]:.1f} us (min {ts[0]:.1f})")
