"""time one staged GEMM launch under both matrix forms (events around 20 launches)"""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from pointcloudlib_amd import _lib
from pointcloudlib_amd.misc.mlp_hip import _P, _stream
dev = torch.device("cuda")
for (rows, cin, cout) in [(4096, 512, 1024), (4096, 256, 512), (32768, 512, 1024), (32768, 128, 1024), (8192, 1024, 512), (131072, 128, 128)]:
    x = torch.randn(rows, cin, device=dev); w = torch.randn(cout, cin, device=dev) / cin ** 0.5
    sc = torch.rand(cin, device=dev) + 0.5; sh = torch.randn(cin, device=dev) * 0.1
    nrows = _lib.size_query("pcl_mlp_stat_rows", rows, cout, 0)
    y = torch.empty(rows, cout, device=dev); stats = torch.empty(nrows, 2, cout, device=dev, dtype=torch.float64)
    out = {}
    for form in (0, 2 | (16 << 8)):
        _lib.lib().pcl_set_matrix_form(form)
        for rep in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                _lib.call("pcl_linear_fwd_rows_f32", _P(x), _P(w), None, _P(sc), _P(sh), 0.0, rows, cin, cout, _P(y), _P(stats), None, None, _stream())
            e1.record(); torch.cuda.synchronize()
        out[form] = e0.elapsed_time(e1) / 20 * 1e3
        ref = torch.relu(x.double() * sc.double() + sh.double()) @ w.double().t()
        err = (y.double() - ref).abs().max().item()
        print(f"rows {rows} {cin}->{cout} form {form:#x}: {out[form]:8.1f} us  {2*rows*cin*cout/out[form]*1e-6:7.1f} TF  max err {err:.2e}")
_lib.lib().pcl_set_matrix_form(0)
