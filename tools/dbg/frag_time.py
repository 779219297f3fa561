"""Timing of the fragment-direct GEMMs (csrc/frag.hip) against the staged kernels (csrc/mlp.hip) on the shapes with few rows:
    python tools/dbg/frag_time.py [P ...]     # default: 4096 (cfg 2 GroupAll level) and 2048 / 8192 (cfg 4: SA3 / fp3, fp2)
us per launch, mean of 40 back-to-back launches after 5 warm-ups (events on the stream), TF = 2 P K N / t."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from pointcloudlib_amd import _lib

dev = torch.device("cuda")
_p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    L = _lib.lib()
    Ps = [int(a) for a in sys.argv[1:]] or [4096, 2048, 8192]
    for P in Ps:
        shapes = [(259, 256), (256, 512), (512, 1024)] if P == 4096 else [(643, 256), (256, 512), (512, 1024), (1664, 256), (256, 256)] if P == 2048 else [(576, 256), (256, 128), (320, 128)]
        print(f"=== P = {P}")
        for K, N in shapes:
            X = torch.randn(P, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5
            sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev)
            Y = torch.empty(P, N, device=dev)
            rows_s = _lib.size_query("pcl_mlp_stat_rows", P, N, 0)
            st_s = torch.empty((max(rows_s, _lib.size_query("pcl_frag_stat_rows", P)), 2, N), dtype=torch.float64, device=dev)
            gf = 2.0 * P * K * N / 1e6
            t0 = timeit(lambda: _lib.call("pcl_linear_fwd_rows_f32", _p(X), _p(W), None, _p(sc), _p(sh), 0.0, P, K, N, _p(Y), _p(st_s), None, None, st()))
            res = [f"staged {t0:6.1f} us ({gf / t0:5.1f} TF)"]
            for flush in (0, 32, 8):
                t1 = timeit(lambda: _lib.call("pcl_frag_linear_fwd_f32", _p(X), K, _p(W), K, None, _p(sc), _p(sh), 0.0, P, K, N, _p(Y), N, None, _p(st_s), flush, st()))
                res.append(f"frag/{flush} {t1:6.1f} us ({gf / t1:5.1f} TF)")
            for tn, ksw in ((1, 1), (1, 2), (1, 4), (2, 1), (2, 2)):
                L.pcl_frag_set_tuning(-1, tn, ksw, 0, 0, 0, 0)
                t1 = timeit(lambda: _lib.call("pcl_frag_linear_fwd_f32", _p(X), K, _p(W), K, None, _p(sc), _p(sh), 0.0, P, K, N, _p(Y), N, None, _p(st_s), 0, st()))
                res.append(f"tn{tn}k{ksw} {t1:5.1f}")
            L.pcl_frag_set_tuning(-1, 0, 0, 0, 0, 0, 0)
            print(f"fwd {K:4d} -> {N:4d}: " + " | ".join(res))
            # ---- backward of the same layer: dy [P, N] ; dX = dy W ; dW = dy^T z
            dU = torch.randn(P, N, device=dev); Yl = torch.randn(P, N, device=dev)
            a, k1, k2, mu = (torch.randn(N, device=dev) for _ in range(4))
            dUp = torch.empty(P, K, device=dev)
            rows_b = _lib.size_query("pcl_mlp_stat_rows", P, K, 1)
            st_b = torch.empty((max(rows_b, _lib.size_query("pcl_frag_stat_rows", P)), 2, K), dtype=torch.float64, device=dev)
            Yp = X
            t0 = timeit(lambda: _lib.call("pcl_linear_bwd_dx_rows_f32", _p(dU), _p(Yl), _p(a), _p(k1), _p(k2), _p(mu), None, None, 1, _p(W), P, N, K, _p(Yp), _p(sc), _p(sh),
                                          0.0, _p(dUp), _p(st_b), None, None, 0, 0, st()))
            dy = torch.empty(P, N, device=dev)
            td = timeit(lambda: _lib.call("pcl_frag_dy_f32", _p(dU), _p(Yl), _p(a), _p(k1), _p(k2), _p(mu), None, None, 1, P, N, _p(dy), None, 0, st()))
            res = [f"staged {t0:6.1f} us ({gf / t0:5.1f} TF)", f"dy pass {td:5.1f} us"]
            t1 = timeit(lambda: _lib.call("pcl_frag_linear_bwd_dx_f32", _p(dy), _p(W), K, P, N, K, _p(Yp), K, _p(sc), _p(sh), 0.0, _p(dUp), K, _p(st_b), 0, st()))
            res.append(f"frag {t1:6.1f} us ({gf / t1:5.1f} TF)")
            for tn, ksw in ((1, 1), (1, 2), (1, 4), (2, 1), (2, 2)):
                L.pcl_frag_set_tuning(-1, tn, ksw, 0, 0, 0, 0)
                t1 = timeit(lambda: _lib.call("pcl_frag_linear_bwd_dx_f32", _p(dy), _p(W), K, P, N, K, _p(Yp), K, _p(sc), _p(sh), 0.0, _p(dUp), K, _p(st_b), 0, st()))
                res.append(f"tn{tn}k{ksw} {t1:5.1f}")
            L.pcl_frag_set_tuning(-1, 0, 0, 0, 0, 0, 0)
            print(f" dx {N:4d} -> {K:4d}: " + " | ".join(res))
            dW = torch.empty(N, K, device=dev)
            nb = _lib.size_query("pcl_linear_bwd_dw_workspace_bytes", P, N, K)
            ws = torch.empty((nb + 3) // 4, device=dev)
            t0 = timeit(lambda: _lib.call("pcl_linear_bwd_dw_rows_f32", _p(dU), _p(Yl), _p(a), _p(k1), _p(k2), _p(mu), None, None, 1, _p(X), _p(sc), _p(sh), 0.0, P, N, K,
                                          _p(dW), _p(ws), nb, None, None, 0, st()))
            res = [f"staged+reduce {t0:6.1f} us ({gf / t0:5.1f} TF)"]
            for shp in ((0, 0, 0, 0), (1, 1, 4, 2), (1, 1, 4, 4), (1, 1, 4, 8), (1, 2, 4, 4), (2, 2, 4, 8), (2, 2, 2, 8), (1, 1, 2, 8)):
                L.pcl_frag_set_tuning(-1, 0, 0, *shp)
                nb2 = L.pcl_frag_dw_workspace_bytes(P, N, K)
                ws2 = torch.zeros(nb2, dtype=torch.uint8, device=dev)
                t1 = timeit(lambda: _lib.call("pcl_frag_linear_bwd_dw_f32", _p(dy), _p(X), K, _p(sc), _p(sh), 0.0, P, N, K, _p(dW), K, _p(ws2), nb2, 1, st()))
                res.append(("auto" if shp == (0, 0, 0, 0) else "t%d%dw%dg%d" % shp) + f" {t1:5.1f}" + (f" ({gf / t1:5.1f} TF)" if shp == (0, 0, 0, 0) else ""))
            L.pcl_frag_set_tuning(-1, 0, 0, 0, 0, 0, 0)
            print(f" dw {N:4d} x {K:4d}: " + " | ".join(res))


if __name__ == "__main__":
    main()
