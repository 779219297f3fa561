"""CPU tests of the oracle itself: hand-derived known answers from the reference's kernel text
 and agreement of the two independent restatements (C thread-emulation vs NumPy
closed forms).  The reference ships no golden vectors, so these pin the oracle as far as it can be."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import np_oracle as npo


def test_optimal_block_natural_log(oracle):
    # misc/ops.py:110-111: 2 ** int(ln B) -> 8 at B=32, 4 at 16 and 8, 2 at 3..7, 1 at <=2
    assert [oracle.optimal_block(b) for b in (1, 2, 3, 7, 8, 16, 20, 21, 32, 54, 55)] == [1, 1, 2, 2, 4, 4, 4, 8, 8, 8, 16]


def test_fps_collinear_hand_trace(oracle):
    # (i): min-dist arrays [0,1,9,49] -> [0,1,9,0] -> [0,1,0,0]  => indices 0,3,2,1
    x = np.array([[[1, 0, 0], [2, 0, 0], [4, 0, 0], [8, 0, 0]]], np.float32)
    for S in (1, 2, 4):
        assert oracle.fps(x, 4, block_size=S).tolist() == [[0, 3, 2, 1]]


def test_fps_tie_rule_bitreverse(oracle):
    # (ii): exact ties go to the smallest (bitreverse_{log2 S}(k mod S), k).
    # k=1..4 are all at squared distance 4 from point 0.  Hand trace for S=4: thread t scans k = t, t+4;
    # thread 0 ends with (4, k=4), threads 1..3 with (4, k=t); the tree keeps the lower tid on ties
    # (misc/ops.py:121) -> thread 0 -> k=4.  Closed form: ranks (2,1),(1,2),(3,3),(0,4) -> k=4.
    x = np.array([[[1, 0, 0], [1, 2, 0], [1, -2, 0], [3, 0, 0], [-1, 0, 0]]], np.float32)
    assert oracle.fps(x, 2, block_size=1)[0, 1] == 1          # S=1: lowest index
    assert oracle.fps(x, 2, block_size=4)[0, 1] == 4
    assert oracle.fps(x, 2, block_size=2)[0, 1] == 2          # S=2: even k first -> (0,2),(0,4),(1,1),(1,3) -> k=2
    # only k=1 and k=2 tied: S=4 -> bitrev2(1)=2 > bitrev2(2)=1 -> k=2 wins (the naive (k mod S, k) rule says k=1)
    y = np.array([[[1, 0, 0], [1, 2, 0], [1, -2, 0], [1.5, 0, 0]]], np.float32)
    assert oracle.fps(y, 2, block_size=4)[0, 1] == 2
    assert oracle.fps(y, 2, block_size=2)[0, 1] == 2
    assert oracle.fps(y, 2, block_size=1)[0, 1] == 1
    for S in (1, 2, 4, 8):
        assert oracle.fps(x, 5, block_size=S).tolist() == npo.fps_np(x, 5, S).tolist()


def test_fps_origin_skip_and_all_skipped(oracle):
    # (iii) a point with squared norm <= 1e-3 is never selected; an all-skipped cloud gives all zeros
    x = np.array([[[0.5, 0, 0], [0.01, 0.01, 0.01], [-0.5, 0, 0], [0, 0.5, 0]]], np.float32)
    idx = oracle.fps(x, 3, block_size=1)
    assert 1 not in idx[0, 1:].tolist()
    tiny = np.full((1, 8, 3), 0.01, np.float32)
    assert oracle.fps(tiny, 5, block_size=2).tolist() == [[0, 0, 0, 0, 0]]
    # threshold is a double compare against 1e-3: mag == float32(1e-3) (> 1e-3) is NOT skipped
    m = np.float32(1e-3)
    x2 = np.array([[[1, 0, 0], [np.sqrt(m), 0, 0], [-1, 0, 0]]], np.float32)
    mag = np.float32(x2[0, 1, 0] * x2[0, 1, 0])
    got = oracle.fps(x2, 3, block_size=1)[0].tolist()
    assert (1 in got) == (float(mag) > 1e-3)


def test_fps_no_skip_and_start_idx(oracle):
    tiny = (np.arange(24, dtype=np.float32).reshape(1, 8, 3)) * 1e-3
    a = oracle.fps(tiny, 4, block_size=1, skip=False, start_idx=np.array([3], np.int32))
    assert a[0, 0] == 3 and len(set(a[0].tolist())) == 4


def test_ball_query_rules(oracle):
    pts = np.array([[[0, 0, 0], [0.1, 0, 0], [0.2, 0, 0], [0.3, 0, 0], [5, 5, 5]]], np.float32)
    q = pts[:, :1]
    # (iv) <ns hits -> padded with first hit; strict '<' on fl(r*r)
    idx, cnt = oracle.ball_query(q, pts, 0.25, 4, return_cnt=True)
    assert idx.tolist() == [[[0, 1, 2, 0]]] and cnt.tolist() == [[3]]
    idx, cnt = oracle.ball_query(q, pts, 0.25, 2, return_cnt=True)          # >= ns hits -> first ns ascending
    assert idx.tolist() == [[[0, 1]]] and cnt.tolist() == [[2]]
    r = np.float32(0.5)
    d = np.sqrt(np.float32(r * r))                                          # d*d == fl(r*r) -> excluded
    pts2 = np.array([[[0, 0, 0], [d, 0, 0]]], np.float32)
    if np.float32(d * d) == np.float32(r * r):
        assert oracle.ball_query(pts2[:, :1], pts2, r, 2).tolist() == [[[0, 0]]]
    far = np.array([[[9, 9, 9]]], np.float32)                               # no hit -> defined as zeros
    idx, cnt = oracle.ball_query(far, pts, 0.1, 3, return_cnt=True)
    assert idx.tolist() == [[[0, 0, 0]]] and cnt.tolist() == [[0]]


def test_group_layout(oracle):
    rng = np.random.default_rng(1)
    xyz = rng.standard_normal((2, 6, 3)).astype(np.float32)
    feat = rng.standard_normal((2, 6, 4)).astype(np.float32)
    new_xyz = xyz[:, :2]
    idx = rng.integers(0, 6, (2, 2, 3)).astype(np.int32)
    out = oracle.group(xyz, new_xyz, feat, idx)                             # (vi) [local_xyz, feat]
    assert out.shape == (2, 2, 3, 7)
    assert np.array_equal(out[1, 1, 2, :3], xyz[1, idx[1, 1, 2]] - new_xyz[1, 1])
    assert np.array_equal(out[1, 1, 2, 3:], feat[1, idx[1, 1, 2]])
    ga = oracle.group_all(xyz, feat)                                        # un-centred
    assert ga.shape == (2, 1, 6, 7) and np.array_equal(ga[0, 0, :, :3], xyz[0])
    assert np.array_equal(out, npo.group_np(xyz, new_xyz, feat, idx))


def test_knn_rules(oracle):
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 5, 12)).astype(np.float32)
    idx = oracle.knn(x, x, 4)                                               # (v) self at rank 0
    assert np.array_equal(idx[:, 0, :], np.tile(np.arange(12), (2, 1)))
    dup = np.zeros((1, 2, 6), np.float32)                                   # all refs identical -> index order
    assert oracle.knn(dup[:, :, :3], dup, 6)[0, :, 0].tolist() == [0, 1, 2, 3, 4, 5]     # k == Nr
    assert np.array_equal(oracle.knn(x[:, :, :7], x, 12), npo.knn_np(x[:, :, :7], x, 12))


def test_three_nn_defined_semantics(oracle):
    rng = np.random.default_rng(3)
    a = rng.standard_normal((2, 9, 3)).astype(np.float32)
    b = a[:, :5]
    idx, w = oracle.three_nn(a, b)
    assert np.array_equal(idx[:, :5, 0], np.tile(np.arange(5), (2, 1)))    # own point first
    assert np.allclose(w.sum(-1), 1, atol=1e-6)
    i2, w2 = npo.three_nn_np(a, b)
    assert np.array_equal(idx, i2) and np.array_equal(w, w2)
    idx1, w1 = oracle.three_nn(a, b[:, :1])                                 # S == 1 broadcast
    assert (idx1 == 0).all() and np.array_equal(w1[..., 0], np.ones((2, 9), np.float32))


# ---- the two restatements agree on tie-heavy random inputs ----------------------------------
@settings(max_examples=60, deadline=None)
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 3), st.integers(2, 48), st.sampled_from([1, 2, 4, 8, 16]),
       st.sampled_from(["lattice", "gauss", "tiny"]))
def test_fps_two_restatements_agree(seed, B, N, S, kind):
    import oracle
    rng = np.random.default_rng(seed)
    if kind == "lattice":
        pts = rng.integers(-2, 3, (B, N, 3)).astype(np.float32) * 0.5
    elif kind == "gauss":
        pts = rng.standard_normal((B, N, 3)).astype(np.float32)
    else:
        pts = rng.integers(-3, 4, (B, N, 3)).astype(np.float32) * 0.01
    m = int(rng.integers(1, N + 1))
    assert np.array_equal(oracle.fps(pts, m, block_size=S), npo.fps_np(pts, m, S))


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 2 ** 31 - 1), st.integers(2, 64), st.integers(1, 24), st.floats(0.05, 1.5))
def test_ball_query_two_restatements_agree(seed, N, ns, radius):
    import oracle
    rng = np.random.default_rng(seed)
    pts = (rng.integers(-4, 5, (2, N, 3)) * 0.125).astype(np.float32)
    m = int(rng.integers(1, N + 1))
    q = pts[:, :m]
    a, ca = oracle.ball_query(q, pts, radius, ns, return_cnt=True)
    b, cb = npo.ball_query_np(q, pts, radius, ns)
    assert np.array_equal(a, b) and np.array_equal(ca, cb)


@settings(max_examples=30, deadline=None)
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 24), st.integers(2, 48), st.integers(1, 40))
def test_knn_two_restatements_agree(seed, C, Nr, Nq):
    import oracle
    rng = np.random.default_rng(seed)
    r = rng.integers(-2, 3, (2, C, Nr)).astype(np.float32)
    q = rng.integers(-2, 3, (2, C, Nq)).astype(np.float32)
    k = int(rng.integers(1, Nr + 1))
    assert np.array_equal(oracle.knn(q, r, k), npo.knn_np(q, r, k))
