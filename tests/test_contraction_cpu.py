"""CPU: the oracle's second reading (nvcc -fmad=true contraction, ``oracle.contract("fma")``) is pinned against exact
rational arithmetic, shown to be a DIFFERENT function from the default reading on constructed inputs, and the
contraction-sensitivity report's bounds (profiles/r03_contraction_sensitivity.txt) are asserted on the
quick variant.  Also: the matmul-form ``knn_point`` restatement (misc/pointconv_utils.py:34-53,:120-131) against an
independent NumPy evaluation."""
import ctypes
from fractions import Fraction

import numpy as np

F = np.float32


def _round_f32(q):
    """exact round-to-nearest-even of a Fraction to float32 (normal range)"""
    if q == 0:
        return F(0)
    sign = -1 if q < 0 else 1
    q = abs(q)
    e = 0
    while q >= 2 ** 24:
        q /= 2; e += 1
    while q < 2 ** 23:
        q *= 2; e -= 1
    n = q.numerator // q.denominator
    rem = q - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (n & 1)):
        n += 1
    return F(sign * float(n) * 2.0 ** e)


def _sumsq3_exact(a, b, c, contract):
    a, b, c = (Fraction(float(v)) for v in (a, b, c))
    if contract:
        t = Fraction(float(_round_f32(a * a)))
        t = Fraction(float(_round_f32(b * b + t)))
        return _round_f32(c * c + t)
    aa, bb, cc = (Fraction(float(_round_f32(v * v))) for v in (a, b, c))
    t = Fraction(float(_round_f32(aa + bb)))
    return _round_f32(t + cc)


def test_sumsq3_both_readings_against_exact_arithmetic(oracle):
    lib = ctypes.CDLL(oracle.lib_path())
    lib.pclo_sumsq3.restype = ctypes.c_float
    lib.pclo_sumsq3.argtypes = [ctypes.c_float] * 3 + [ctypes.c_int]
    rng = np.random.default_rng(0)
    differ = 0
    for a, b, c in rng.standard_normal((3000, 3)).astype(F):
        r0, r1 = lib.pclo_sumsq3(a, b, c, 0), lib.pclo_sumsq3(a, b, c, 1)
        assert F(r0) == _sumsq3_exact(a, b, c, 0) and F(r1) == _sumsq3_exact(a, b, c, 1)
        differ += r0 != r1
    assert differ > 300          # the two readings disagree in the last place on a large share of random triples


def test_knn_readings_order_a_constructed_near_tie_differently(oracle):
    """two references whose distances to the query are ordered one way uncontracted and the other way contracted"""
    rng = np.random.default_rng(1)
    q = np.array([0.1, -0.2, 0.3], F)
    d = rng.standard_normal((20000, 3))
    refs = (q + d / np.linalg.norm(d, axis=1, keepdims=True) * (1 + rng.uniform(-2e-7, 2e-7, (20000, 1)))).astype(F)
    lib = ctypes.CDLL(oracle.lib_path())
    lib.pclo_sumsq3.restype = ctypes.c_float
    lib.pclo_sumsq3.argtypes = [ctypes.c_float] * 3 + [ctypes.c_int]
    t = refs - q                                                      # misc/ops.py:489 tmp = ref - qry (fp32)
    d0 = np.array([lib.pclo_sumsq3(*r, 0) for r in t[:4000]], F)
    d1 = np.array([lib.pclo_sumsq3(*r, 1) for r in t[:4000]], F)
    # kNN accumulates from ssd = 0: ((0 + a*a) + b*b) + c*c == sumsq3 mode 0; fma(c,c,fma(b,b,fma(a,a,0))) == mode 1
    pair = None
    for i in range(4000):
        j = np.flatnonzero((d0[i] < d0) & (d1[i] > d1))
        if len(j):
            pair = (i, int(j[0])); break
    assert pair is not None
    i, j = pair
    assert _sumsq3_exact(*t[i], 0) < _sumsq3_exact(*t[j], 0) and _sumsq3_exact(*t[i], 1) > _sumsq3_exact(*t[j], 1)
    x_r = np.ascontiguousarray(refs[[i, j]].T[None])                  # [1,3,2]
    x_q = np.ascontiguousarray(q[None, :, None])                      # [1,3,1]
    assert oracle.get_contract() == 0
    assert oracle.knn(x_q, x_r, 1)[0, 0, 0] == 0
    with oracle.contract("fma"):
        assert oracle.get_contract() == 1
        assert oracle.knn(x_q, x_r, 1)[0, 0, 0] == 1
    assert oracle.get_contract() == 0
    # ball query on the same pair: a radius^2 that separates the two readings for reference i
    lo, hi = sorted((float(d0[i]), float(d1[i])))
    if lo != hi:
        r = F(np.sqrt((lo + hi) / 2))
        r2 = F(r * r)
        if lo < r2 <= hi:
            pts = refs[[i]][None]
            a = oracle.ball_query(q[None, None], pts, r, 2, return_cnt=True)[1][0, 0]
            with oracle.contract("fma"):
                b = oracle.ball_query(q[None, None], pts, r, 2, return_cnt=True)[1][0, 0]
            assert {int(a), int(b)} == {0, 1}


def test_fps_fma_reading_matches_numpy_emulation(oracle):
    """FPS under the contracted reading == a NumPy FPS whose distances come from the exact-arithmetic emulation"""
    rng = np.random.default_rng(2)
    pts = rng.standard_normal((2, 40, 3)).astype(F) * F(0.5)
    with oracle.contract("fma"):
        got = oracle.fps(pts, 12, block_size=1, skip=False)
    for b in range(2):
        temp = np.full(40, F(1e10), F)
        old, seq = 0, [0]
        for _ in range(11):
            d = np.array([_sumsq3_exact(*(pts[b, k] - pts[b, old]), 1) for k in range(40)], F)
            temp = np.minimum(d, temp)
            old = int(np.argmax(temp))
            seq.append(old)
        assert got[b].tolist() == seq


def test_knn_point_matmul_matches_numpy(oracle):
    rng = np.random.default_rng(3)
    xyz = rng.standard_normal((2, 96, 3)).astype(F)
    q = xyz[:, ::3].copy()
    idx, dist = oracle.knn_point_matmul(8, xyz, q, fma_dot=False, return_dist=True)
    dot = (q[:, :, None, 0] * xyz[:, None, :, 0] + q[:, :, None, 1] * xyz[:, None, :, 1]) + q[:, :, None, 2] * xyz[:, None, :, 2]
    want = F(-2) * dot
    want = want + ((q[..., 0] ** 2 + q[..., 1] ** 2) + q[..., 2] ** 2)[:, :, None]
    want = want + ((xyz[..., 0] ** 2 + xyz[..., 1] ** 2) + xyz[..., 2] ** 2)[:, None, :]
    assert np.array_equal(dist, want)
    assert np.array_equal(idx, np.argsort(want, axis=-1, kind="stable")[..., :8])
    # a query that IS a reference point need not come back at rank 0 in matmul form (cancellation), but its
    # matmul-form distance is within a few ulps of |a|^2 of zero
    self_d = dist[:, np.arange(32), np.arange(0, 96, 3)]
    assert np.abs(self_d).max() < 1e-5


def test_contraction_sensitivity_bounds_quick():
    """the numbers DESIGN.md quotes from profiles/r03_contraction_sensitivity.txt, re-derived on the quick variant"""
    from contraction_sensitivity import run
    lines, st = run(quick=True)
    for k, v in st.items():
        if "fps" in k:
            assert v["clouds"] <= 0.25, (k, v)            # full-size report: 0 of 32 / 16 clouds on every config
        elif "bq" in k:
            assert v["rows"] <= 2e-3, (k, v)
        elif "knn" in k:
            assert v["sets"] <= 1e-3 and v["rows"] <= 5e-3, (k, v)
        elif "mm" in k:
            assert v["sets"] <= 5e-3 and v["rows"] <= 1e-2, (k, v)
