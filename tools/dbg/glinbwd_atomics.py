"""The scatter of the folded first layer's backward at the SA2 shape of PointNet++ SSG cls (176 k compacted rows x 128 channels into 32 x 512
points): fp32 atomics (pcl_group_linear_bwd_f32, with and without the scatter target) against the gather over the points' row lists
(pcl_group_rows_transpose_i32 + pcl_group_linear_bwd_gather_f32), which is also checked against the atomic form and against torch.
    python tools/dbg/glinbwd_atomics.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from pointcloudlib_amd import _lib

dev = torch.device("cuda")
_p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B, N, m, ns, C1 = 32, 512, 128, 64, 128
torch.manual_seed(0)
rows_blk = _lib.lib().pcl_group_linear_stat_rows(B, m)
# ball-query-like groups: cnt distinct rows per group, sources = random points of the group's cloud
cnt = torch.randint(20, 64, (B * m,), device=dev)
goff = torch.zeros(B * m + 1, dtype=torch.int32, device=dev); goff[1:] = torch.cumsum(cnt, 0).int()
R = int(goff[-1].item())
grp = torch.repeat_interleave(torch.arange(B * m, device=dev), cnt)
score = torch.rand(B * m, N, device=dev)
score[:, :8] = -1.0                     # popular points: the first 8 points of a cloud are in every group (like the centre of a gauss_ball cloud)
pick = score.argsort(1)[:, :ns]         # distinct sources within a group
slot = torch.arange(R, device=dev) - goff[:-1].long()[grp]
src = ((grp // m) * N + pick[grp, slot]).int()
loc = torch.randn(R, 4, device=dev); loc[:, 3] = torch.randint(1, 4, (R,), device=dev).float()
dU, Y = torch.randn(R, C1, device=dev), torch.randn(R, C1, device=dev)
a, k1, k2, mu = (torch.randn(C1, device=dev) for _ in range(4))
nrows = goff[B * m:]
dUf = torch.empty(B * N, C1, device=dev)
dWx = torch.empty(rows_blk, C1, 3, device=dev)
in_off = torch.empty(B * N + 1, dtype=torch.int32, device=dev)
in_rows = torch.empty(R, dtype=torch.int32, device=dev)

def atomics(with_scatter=True):
    _lib.call("pcl_group_linear_bwd_f32", _p(loc), None, 0, _p(dU), _p(Y), _p(a), _p(k1), _p(k2), _p(mu), _p(src), _p(nrows), B, N, C1,
              _p(dUf) if with_scatter else None, _p(dWx), None, None, 0, 0, st())

def transpose():
    _lib.call("pcl_group_rows_transpose_i32", _p(src), _p(goff), B, N, m, ns, _p(in_off), _p(in_rows), st())

def gather():
    _lib.call("pcl_group_linear_bwd_gather_f32", _p(loc), _p(dU), _p(Y), _p(a), _p(k1), _p(k2), _p(mu), _p(in_off), _p(in_rows), B, N, C1, _p(dUf), _p(dWx),
              None, 0, st())

# ---- correctness
transpose(); torch.cuda.synchronize()
order = torch.sort(src.long() * (R + 1) + torch.arange(R, device=dev)).indices.int()          # rows by (source, row)
counts = torch.bincount(src.long(), minlength=B * N)
ref_off = torch.zeros(B * N + 1, dtype=torch.int64, device=dev); ref_off[1:] = torch.cumsum(counts, 0)
assert torch.equal(in_off.long(), ref_off), "in_off"
assert torch.equal(in_rows, order), "in_rows"
dy = a * dU - loc[:, 3:4] * (k1 + k2 * (Y - mu))
ref = torch.zeros(B * N, C1, device=dev, dtype=torch.float64).index_add_(0, src.long(), dy.double())
atomics(); torch.cuda.synchronize(); ua, wa = dUf.clone(), dWx.sum(0)
gather(); torch.cuda.synchronize(); ug, wg = dUf.clone(), dWx.sum(0)
gather(); torch.cuda.synchronize()
assert torch.equal(ug, dUf), "the gather is run-to-run identical"
sc = ref.abs().max().item()
print(f"max|dUf - fp64| / max|dUf|: atomics {(ua.double() - ref).abs().max().item() / sc:.2e}, gather {(ug.double() - ref).abs().max().item() / sc:.2e}; "
      f"dWx partial sums, gather vs atomics kernel: {(wa - wg).abs().max().item() / wa.abs().max().item():.2e}")

def timeit(fn, n=40):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

print(f"atomics (incl. the memset of dUf) {timeit(atomics):7.1f} us | without the scatter target {timeit(lambda: atomics(False)):7.1f} us | "
      f"gather {timeit(gather):7.1f} us | transpose (once per forward) {timeit(transpose):7.1f} us   [R = {R} rows]")
