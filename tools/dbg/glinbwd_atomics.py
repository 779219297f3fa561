"""What do the float atomics of pcl_group_linear_bwd_f32 cost?  The SA2 shape of PointNet++ SSG cls (176 k compacted rows x 128 channels scattered to
32 x 512 points) with and without the scatter target (dUf = None: only the dWx partial sums).   python tools/dbg/glinbwd_atomics.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from pointcloudlib_amd import _lib

dev = torch.device("cuda")
_p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B, N, m, ns, C1 = 32, 512, 128, 64, 128
R = 176252
torch.manual_seed(0)
rows_blk = _lib.lib().pcl_group_linear_stat_rows(B, m)
# rows of group g come from cloud g // m: ball-query-like locality (a group's rows are points near its centre: here random points of its cloud)
grp = torch.sort(torch.randint(0, B * m, (R,), device=dev)).values
src = (grp // m) * N + torch.randint(0, N, (R,), device=dev)
src = src.int()
loc = torch.randn(R, 4, device=dev); loc[:, 3] = 1.0
dU, Y = torch.randn(R, C1, device=dev), torch.randn(R, C1, device=dev)
a, k1, k2, mu = (torch.randn(C1, device=dev) for _ in range(4))
nrows = torch.tensor([R], dtype=torch.int32, device=dev)
dUf = torch.empty(B * N, C1, device=dev)
dWx = torch.empty(rows_blk, C1, 3, device=dev)

def run(with_scatter):
    _lib.call("pcl_group_linear_bwd_f32", _p(loc), None, 0, _p(dU), _p(Y), _p(a), _p(k1), _p(k2), _p(mu), _p(src), _p(nrows), B, N, C1,
              _p(dUf) if with_scatter else None, _p(dWx), None, None, 0, 0, st())

for ws in (True, False, True, False):
    for _ in range(5):
        run(ws)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(40):
        run(ws)
    e1.record(); torch.cuda.synchronize()
    print(f"scatter to dUf {'on ' if ws else 'off'}: {e0.elapsed_time(e1) / 40 * 1e3:7.1f} us per call (incl. the memset of dUf when on)")
