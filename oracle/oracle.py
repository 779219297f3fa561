"""ctypes front-end of ``pcl_oracle.c`` (NumPy in, NumPy out).  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_float_p = ctypes.POINTER(ctypes.c_float)
c_int_p = ctypes.POINTER(ctypes.c_int32)


def lib_path():
    return os.path.join(_HERE, "_build", "libpcl_oracle.so")


def build(force=False):
    """Compile the C restatement with gcc (``make -C oracle``)."""
    src = os.path.join(_HERE, "pcl_oracle.c")
    out = lib_path()
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)
    return out


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(c_float_p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(c_int_p)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed rc={rc}")


def num_threads():
    return int(_lib().pclo_num_threads())


def optimal_block(batch_size):
    return int(_lib().pclo_optimal_block(int(batch_size)))


def fps(xyz, m, block_size=None, skip=True, start_idx=None, return_xyz=False):
    """xyz [B,N,3] -> idx int32 [B,m] (and gathered xyz [B,m,3])."""
    xyz, px = _f(xyz)
    B, N, _ = xyz.shape
    if block_size is None:
        block_size = optimal_block(B)
    idx = np.empty((B, m), np.int32)
    new_xyz = np.empty((B, m, 3), np.float32)
    ps = None
    if start_idx is not None:
        start_idx, ps = _i(start_idx)
    _check(_lib().pclo_fps_f32(px, B, N, m, int(block_size), int(bool(skip)), ps,
                               idx.ctypes.data_as(c_int_p), new_xyz.ctypes.data_as(c_float_p)), "fps")
    return (idx, new_xyz) if return_xyz else idx


def ball_query(new_xyz, xyz, radius, nsample, return_cnt=False):
    new_xyz, pq = _f(new_xyz)
    xyz, pp = _f(xyz)
    B, m, _ = new_xyz.shape
    N = xyz.shape[1]
    idx = np.empty((B, m, nsample), np.int32)
    cnt = np.empty((B, m), np.int32)
    _check(_lib().pclo_ball_query_f32(pq, pp, B, m, N, ctypes.c_float(np.float32(radius)), int(nsample),
                                      idx.ctypes.data_as(c_int_p), cnt.ctypes.data_as(c_int_p)), "ball_query")
    return (idx, cnt) if return_cnt else idx


def group(xyz, new_xyz, feat, idx, use_xyz=True):
    idx, pi = _i(idx)
    B, m, ns = idx.shape
    xyz, px = _f(xyz)
    new_xyz, pq = _f(new_xyz)
    N = xyz.shape[1]
    C = 0
    pf = None
    if feat is not None:
        feat, pf = _f(feat)
        C = feat.shape[2]
    D = (3 if use_xyz else 0) + C
    out = np.empty((B, m, ns, D), np.float32)
    _check(_lib().pclo_group_f32(px, pq, pf, pi, B, N, m, ns, C, int(bool(use_xyz)),
                                 out.ctypes.data_as(c_float_p)), "group")
    return out


def group_bwd(gout, idx, N, C, use_xyz=True):
    gout, pg = _f(gout)
    idx, pi = _i(idx)
    B, m, ns = idx.shape
    gfeat = np.empty((B, N, C), np.float32)
    _check(_lib().pclo_group_bwd_f32(pg, pi, B, N, m, ns, C, int(bool(use_xyz)),
                                     gfeat.ctypes.data_as(c_float_p)), "group_bwd")
    return gfeat


def group_all(xyz, feat, use_xyz=True):
    xyz, px = _f(xyz)
    B, N, _ = xyz.shape
    C = 0
    pf = None
    if feat is not None:
        feat, pf = _f(feat)
        C = feat.shape[2]
    D = (3 if use_xyz else 0) + C
    out = np.empty((B, 1, N, D), np.float32)
    _check(_lib().pclo_group_all_f32(px, pf, B, N, C, int(bool(use_xyz)), out.ctypes.data_as(c_float_p)),
           "group_all")
    return out


def knn(x_q, x_r, k):
    """KNN(k).execute(x_q[B,C,Nq], x_r[B,C,Nr]) -> int32 [B,k,Nq]  (misc/ops.py:651-663)."""
    x_q, pq = _f(x_q)
    x_r, pr = _f(x_r)
    B, C, Nq = x_q.shape
    Nr = x_r.shape[2]
    idx = np.empty((B, k, Nq), np.int32)
    _check(_lib().pclo_knn_f32(pr, pq, B, C, Nr, Nq, int(k), idx.ctypes.data_as(c_int_p), None), "knn")
    return idx


def set_contract(mode):
    """0: every fp32 operation of the index-producing distances rounded on its own (the library's definition);
    1 / "fma": nvcc's default -fmad=true reading (see pcl_oracle.c header).  Process-global."""
    _lib().pclo_set_contract(1 if mode in (1, True, "fma") else 0)


def get_contract():
    return int(_lib().pclo_get_contract())


class contract:
    """``with oracle.contract("fma"): ...`` -- evaluate the oracle under the contracted reading."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = get_contract()
        set_contract(self.mode)

    def __exit__(self, *a):
        set_contract(self.prev)


def knn_point_matmul(nsample, xyz, new_xyz, fma_dot=True, return_dist=False):
    """misc/pointconv_utils.py:120-131 in the reference's matmul-form arithmetic: xyz [B,N,3], new_xyz [B,S,3] ->
    int32 [B,S,nsample] (stable ascending argsort of -2ab + a^2 + b^2)."""
    xyz, px = _f(xyz)
    new_xyz, pq = _f(new_xyz)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    idx = np.empty((B, S, nsample), np.int32)
    dist = np.empty((B, S, N), np.float32) if return_dist else None
    _check(_lib().pclo_knn_point_matmul_f32(px, pq, B, N, S, int(nsample), int(bool(fma_dot)), idx.ctypes.data_as(c_int_p),
                                            dist.ctypes.data_as(c_float_p) if return_dist else None), "knn_point_matmul")
    return (idx, dist) if return_dist else idx


def three_nn(xyz1, xyz2):
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    idx = np.empty((B, N, 3), np.int32)
    w = np.empty((B, N, 3), np.float32)
    _check(_lib().pclo_three_nn_f32(p1, p2, B, N, S, idx.ctypes.data_as(c_int_p), w.ctypes.data_as(c_float_p)),
           "three_nn")
    return idx, w


def three_interp(points2, idx3, w3):
    points2, pp = _f(points2)
    idx3, pi = _i(idx3)
    w3, pw = _f(w3)
    B, S, D = points2.shape
    N = idx3.shape[1]
    out = np.empty((B, N, D), np.float32)
    _check(_lib().pclo_three_interp_f32(pp, pi, pw, B, N, S, D, out.ctypes.data_as(c_float_p)), "three_interp")
    return out


def density(xyz, bandwidth):
    """PointConv compute_density (misc/pointconv_utils.py:174-184), direct-form distances, fp64 accumulation."""
    xyz, px = _f(xyz)
    B, N, _ = xyz.shape
    out = np.empty((B, N), np.float32)
    _check(_lib().pclo_density_f32(px, B, N, ctypes.c_float(bandwidth), out.ctypes.data_as(c_float_p)), "density")
    return out
