"""GPU parity tests: every HIP kernel, called through the C ABI, against the CPU oracle on the same
seeded inputs -- bit-exact for indices, and for pure copies/subtractions; tolerance stated where fp32 sums
are reordered (atomics).  Plus the committed fixtures and full-size digests."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from pointcloudlib_amd import synth
from pointcloudlib_amd.misc import ops

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


# ------------------------------------------------------------------------------------ FPS
@pytest.mark.parametrize("B,N,m", [(1, 64, 64), (2, 100, 37), (3, 512, 128), (16, 1024, 512), (32, 1024, 512),
                                   (4, 2048, 512), (2, 4096, 1024), (1, 5000, 300), (1, 12000, 64), (1, 20000, 16)])
def test_fps_matches_oracle(oracle, dev, B, N, m):
    pts = synth.gauss_ball(B, N, 100 + N + B)
    for S in (1, oracle.optimal_block(B)):
        want, want_xyz = oracle.fps(pts, m, block_size=S, return_xyz=True)
        idx, new_xyz = ops.furthest_point_sample(T(pts, dev), m, tie_stride=S)
        assert np.array_equal(idx.cpu().numpy(), want), (B, N, m, S)
        assert np.array_equal(new_xyz.cpu().numpy(), want_xyz)


@pytest.mark.parametrize("threads", [64, 128, 256, 512, 1024])
def test_fps_every_launch_shape(oracle, dev, threads):
    """The winner must not depend on the workgroup shape (tie rule is an explicit parameter) nor on the waves' issue priority
    (pcl_set_fps_tuning: a process-wide setting made between calls; it replaced an environment variable read inside pcl_fps_f32)."""
    from pointcloudlib_amd import _lib
    try:
        _lib.lib().pcl_set_fps_tuning(threads, threads % 3)
        for name, pts in synth.adversarial_clouds(0).items():
            B, N, _ = pts.shape
            for S in (1, 2, 4, 8):
                want = oracle.fps(pts, N // 2, block_size=S)
                got, _ = ops.furthest_point_sample(torch.from_numpy(pts).cuda(), N // 2, tie_stride=S)
                assert np.array_equal(got.cpu().numpy(), want), (name, S)
        pts = synth.gauss_ball(4, 1024, 5)
        want = oracle.fps(pts, 256, block_size=4)
        got, _ = ops.furthest_point_sample(torch.from_numpy(pts).cuda(), 256, tie_stride=4)
        assert np.array_equal(got.cpu().numpy(), want)
    finally:
        _lib.lib().pcl_set_fps_tuning(0, 3)


def test_fps_adversarial_ties_and_skips(oracle, dev):
    g = np.load(os.path.join(GOLD, "small_cases.npz"))
    for name, pts in synth.adversarial_clouds(0).items():
        N = pts.shape[1]
        m = max(2, N // 3)
        for S in (1, 2, 4, 8, 16):
            idx, _ = ops.furthest_point_sample(T(pts, dev), m, tie_stride=S)
            assert np.array_equal(idx.cpu().numpy(), g[f"{name}.fps_S{S}"]), (name, S)
            assert np.array_equal(idx.cpu().numpy(), oracle.fps(pts, m, block_size=S))


def test_fps_no_skip_start_idx(oracle, dev):
    pts = synth.gauss_ball(5, 300, 9) * 0.02             # everything near the origin
    start = np.array([0, 7, 299, 150, 3], np.int32)
    want = oracle.fps(pts, 64, block_size=1, skip=False, start_idx=start)
    idx, _ = ops.furthest_point_sample(T(pts, dev), 64, tie_stride=1, skip_sqnorm_le=None, start_idx=T(start, dev))
    assert np.array_equal(idx.cpu().numpy(), want)
    # with the skip enabled the whole cloud is dead: all zeros (misc/ops.py:152-153)
    idx, _ = ops.furthest_point_sample(T(pts, dev), 8, tie_stride=1)
    assert np.array_equal(idx.cpu().numpy(), oracle.fps(pts, 8, block_size=1))


def test_fps_properties_full_size(dev):
    """Size-independent properties at BASELINE size: distinct indices, farthest-first radii non-increasing."""
    pts = synth.gauss_ball(32, 4096, 77)
    idx, new_xyz = ops.furthest_point_sample(T(pts, dev), 1024)
    idx = idx.cpu().numpy()
    live = (pts.astype(np.float32) ** 2).sum(-1) > 1e-3
    for b in range(32):
        assert len(set(idx[b].tolist())) == 1024
        assert live[b, idx[b, 1:]].all()
    x = new_xyz.cpu().numpy()[0].astype(np.float64)
    d = np.full(1024, np.inf)
    radii = []
    mask = live[0]
    P = pts[0].astype(np.float64)
    run = np.full(4096, np.inf)
    for j in range(1, 200):
        run = np.minimum(run, ((P - x[j - 1]) ** 2).sum(-1))
        radii.append(run[mask].max())
    assert all(radii[i] >= radii[i + 1] - 1e-12 for i in range(len(radii) - 1))


# ------------------------------------------------------------------------------------ ball query
@pytest.mark.parametrize("B,N,m,r,ns", [(2, 64, 10, 0.3, 8), (3, 1000, 77, 0.2, 64), (32, 1024, 512, 0.2, 64),
                                        (32, 512, 128, 0.4, 64), (2, 4096, 512, 0.1, 16), (2, 2048, 512, 0.4, 128),
                                        (1, 14000, 33, 0.15, 32)])
def test_ball_query_matches_oracle(oracle, dev, B, N, m, r, ns):
    pts = synth.gauss_ball(B, N, 11 + N)
    q = oracle.fps(pts, m, block_size=1, return_xyz=True)[1]
    want, wcnt = oracle.ball_query(q, pts, r, ns, return_cnt=True)
    idx, cnt = ops.ball_query(T(q, dev), T(pts, dev), r, ns, return_cnt=True)
    assert np.array_equal(idx.cpu().numpy(), want)
    assert np.array_equal(cnt.cpu().numpy(), wcnt)


def test_ball_query_edges(oracle, dev):
    g = np.load(os.path.join(GOLD, "small_cases.npz"))
    for name, pts in synth.adversarial_clouds(0).items():
        m = max(2, pts.shape[1] // 3)
        q = oracle.fps(pts, m, block_size=1, return_xyz=True)[1]
        for r, ns in ((0.1, 4), (0.3, 8), (1.0, 16)):
            idx, cnt = ops.ball_query(T(q, dev), T(pts, dev), r, ns, return_cnt=True)
            assert np.array_equal(idx.cpu().numpy(), g[f"{name}.bq_r{r}_ns{ns}"]), (name, r, ns)
            assert np.array_equal(cnt.cpu().numpy(), g[f"{name}.bqcnt_r{r}_ns{ns}"])
    # queries that are not cloud points: rows with no hit are zero-filled, cnt 0
    pts = synth.sphere_shell(2, 256, 3)
    q = np.zeros((2, 4, 3), np.float32) + 5.0
    idx, cnt = ops.ball_query(T(q, dev), T(pts, dev), 0.5, 8, return_cnt=True)
    assert (idx.cpu().numpy() == 0).all() and (cnt.cpu().numpy() == 0).all()
    # shell clouds never saturate: every query is a full scan with padding
    q = oracle.fps(pts, 64, block_size=1, return_xyz=True)[1]
    assert np.array_equal(ops.ball_query(T(q, dev), T(pts, dev), 0.2, 64).cpu().numpy(), oracle.ball_query(q, pts, 0.2, 64))


# ------------------------------------------------------------------------------------ grouping
@pytest.mark.parametrize("C,use_xyz", [(3, True), (128, True), (0, True), (5, False)])
def test_group_fwd_bwd(oracle, dev, C, use_xyz):
    B, N, m, ns = 4, 300, 50, 16
    rng = np.random.default_rng(C)
    pts = synth.gauss_ball(B, N, 21)
    feat = rng.standard_normal((B, N, C)).astype(np.float32) if C else None
    q = oracle.fps(pts, m, block_size=1, return_xyz=True)[1]
    idx = oracle.ball_query(q, pts, 0.3, ns)
    want = oracle.group(pts, q, feat, idx, use_xyz)
    f = T(feat, dev).requires_grad_(True) if C else None
    out = ops.group_points(T(pts, dev), T(q, dev), f, T(idx, dev), use_xyz)
    assert np.array_equal(out.detach().cpu().numpy(), want)          # copies and one subtraction: bit-exact
    if C:
        gout = rng.standard_normal(want.shape).astype(np.float32)
        out.backward(T(gout, dev))
        wantg = oracle.group_bwd(gout, idx, N, C, use_xyz)
        # scatter-add via fp32 atomics: order differs from the sequential oracle -> tolerance, not bits
        np.testing.assert_allclose(f.grad.cpu().numpy(), wantg, rtol=1e-5, atol=1e-5)


def test_group_all_and_modules(oracle, dev):
    B, N, C = 3, 128, 7
    rng = np.random.default_rng(0)
    pts = synth.gauss_ball(B, N, 5)
    feat = rng.standard_normal((B, N, C)).astype(np.float32)
    f = T(feat, dev).requires_grad_(True)
    out = ops.GroupAll(True)(None, T(pts, dev), f)
    assert np.array_equal(out.detach().cpu().numpy(), oracle.group_all(pts, feat))
    g = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(T(g, dev))
    assert np.array_equal(f.grad.cpu().numpy(), g[:, 0, :, 3:])
    # module signatures of the reference
    y, idx = ops.FurthestPointSampler(32)(T(pts, dev), return_idx=True)
    assert y.shape == (B, 32, 3) and idx.dtype == torch.int32
    grouped = ops.BallQueryGrouper(0.3, 8, True).execute(y, T(pts, dev), T(feat, dev))
    assert grouped.shape == (B, 32, 8, 3 + C)
    want = oracle.group(pts, y.cpu().numpy(), feat, oracle.ball_query(y.cpu().numpy(), pts, 0.3, 8))
    assert np.array_equal(grouped.cpu().numpy(), want)
    assert ops.BallQueryGrouper(0.3, 8, True)(y, T(pts, dev), None).shape == (B, 32, 8, 3)
    gi = ops.index_points(T(feat, dev), idx)
    assert np.array_equal(gi.cpu().numpy(), feat[np.arange(B)[:, None], idx.cpu().numpy()])


# ------------------------------------------------------------------------------------ KNN
@pytest.mark.parametrize("B,C,Nr,Nq,k", [(2, 3, 64, 64, 5), (2, 5, 33, 20, 33), (4, 3, 1024, 1024, 20),
                                         (2, 64, 1024, 1024, 20), (2, 128, 256, 1024, 200), (1, 7, 2048, 100, 40),
                                         (1, 3, 5000, 70, 9),                                 # > 4096 references: two-pass form
                                         (1, 4, 4096, 37, 17), (2, 6, 513, 77, 30), (1, 130, 300, 50, 7),   # fused: 64 values per
                                         (3, 64, 1000, 1000, 20)])    # lane; odd Nr (scalar staging); channel tail; Nr % 256 != 0
def test_knn_matches_oracle(oracle, dev, B, C, Nr, Nq, k):
    rng = np.random.default_rng(Nr + C)
    r = rng.standard_normal((B, C, Nr)).astype(np.float32)
    q = r[:, :, :Nq].copy() if Nq <= Nr else rng.standard_normal((B, C, Nq)).astype(np.float32)
    want = oracle.knn(q, r, k)
    got = ops.KNN(k)(T(q, dev), T(r, dev))
    assert got.shape == (B, k, Nq) and got.dtype == torch.int32
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("B,C,Nr,Nq,k", [(2, 3, 1024, 1024, 20), (2, 64, 1024, 1024, 20), (2, 128, 1024, 512, 20), (1, 64, 2048, 300, 40),
                                         (1, 9, 5000, 77, 16), (1, 128, 256, 256, 200), (3, 7, 333, 100, 5)])
def test_knn_fma_definition_matches_oracle_fma_reading(oracle, dev, B, C, Nr, Nq, k):
    """pcl_knn_fma_f32 -- the named second definition (ssd = fma(tmp, tmp, ssd), nvcc's default contraction of misc/ops.py:490)
    -- is bit-exact against the oracle evaluated under the same reading, fused and two-pass (> 4096 references) paths."""
    rng = np.random.default_rng(Nr + C)
    r = rng.standard_normal((B, C, Nr)).astype(np.float32)
    q = r[:, :, :Nq].copy() if Nq <= Nr else rng.standard_normal((B, C, Nq)).astype(np.float32)
    with oracle.contract("fma"):
        want = oracle.knn(q, r, k)
    got = ops.knn_indices(T(q, dev), T(r, dev), k, contract="fma")
    assert np.array_equal(got.cpu().numpy(), want)
    plain = ops.knn_indices(T(q, dev), T(r, dev), k)                     # and the default stays the source reading
    assert np.array_equal(plain.cpu().numpy(), oracle.knn(q, r, k))


def test_knn_fma_and_default_definitions_differ_on_a_constructed_near_tie(oracle, dev):
    """the two definitions are different functions: a pair of references ordered one way under one, the other way under the other"""
    rng = np.random.default_rng(1)
    q = np.array([0.1, -0.2, 0.3], np.float32)
    d = rng.standard_normal((4000, 3))
    refs = (q + d / np.linalg.norm(d, axis=1, keepdims=True) * (1 + rng.uniform(-2e-7, 2e-7, (4000, 1)))).astype(np.float32)
    x_r = np.ascontiguousarray(refs.T[None])
    x_q = np.ascontiguousarray(q[None, :, None])
    a = oracle.knn(x_q, x_r, 200)[0, :, 0]
    with oracle.contract("fma"):
        b = oracle.knn(x_q, x_r, 200)[0, :, 0]
    assert not np.array_equal(a, b)                                     # (2 ulp spread over 4000 points at one distance: many near-ties)
    assert np.array_equal(ops.knn_indices(T(x_q, dev), T(x_r, dev), 200).cpu().numpy()[0, :, 0], a)
    assert np.array_equal(ops.knn_indices(T(x_q, dev), T(x_r, dev), 200, contract="fma").cpu().numpy()[0, :, 0], b)


def test_knn_ties_by_index(oracle, dev):
    g = np.load(os.path.join(GOLD, "small_cases.npz"))
    for k in (1, 7, 33):
        got = ops.knn_indices(T(g["knn.q"], dev), T(g["knn.r"], dev), k)
        assert np.array_equal(got.cpu().numpy(), g[f"knn.k{k}"])
    z = np.zeros((1, 4, 100), np.float32)
    assert np.array_equal(ops.knn_indices(T(z[:, :, :3], dev), T(z, dev), 100).cpu().numpy()[0, :, 0], np.arange(100))


# ------------------------------------------------------------------------------------ 3-NN interpolation
@pytest.mark.parametrize("N,S", [(200, 40), (2048, 512), (512, 128), (128, 1), (64, 2), (300, 5000)])
def test_three_nn_interp(oracle, dev, N, S):
    B, D = 3, 19
    rng = np.random.default_rng(N + S)
    a = synth.gauss_ball(B, N, 1)
    b = synth.gauss_ball(B, S, 2)
    p2 = rng.standard_normal((B, S, D)).astype(np.float32)
    wi, ww = oracle.three_nn(a, b)
    idx, w = ops.three_nn(T(a, dev), T(b, dev))
    assert np.array_equal(idx.cpu().numpy(), wi)
    assert np.array_equal(w.cpu().numpy(), ww)
    p = T(p2, dev).requires_grad_(True)
    out = ops.three_interpolate(p, idx, w)
    assert np.array_equal(out.detach().cpu().numpy(), oracle.three_interp(p2, wi, ww))
    g = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(T(g, dev))
    ref = torch.zeros(B, S, D, dtype=torch.float64)
    for j in range(3):
        ref.scatter_add_(1, torch.from_numpy(wi[..., j].astype(np.int64))[..., None].expand(B, N, D),
                         torch.from_numpy(g).double() * torch.from_numpy(ww[..., j:j + 1]).double())
    np.testing.assert_allclose(p.grad.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("N,S", [(100, 27), (33, 3), (257, 64), (40, 9000)])
def test_three_nn_ties_across_the_lane_split(oracle, dev, N, S):
    """Eight lanes per target each scan every eighth source and the eight top-3 lists are merged by (distance, index): on clouds made of
    repeated points and lattice points -- equal distances at indices that fall to different lanes -- the lists still equal the oracle's
    sequential scan (the lower index first), bit for bit, weights included."""
    B = 2
    rng = np.random.default_rng(N * S)
    lattice = rng.integers(-2, 3, size=(B, S, 3)).astype(np.float32) * 0.25          # 125 distinct sites: many exact ties, many repeats
    lattice[:, S // 2:] = lattice[:, : S - S // 2]                                    # ... and every site at least twice
    tgt = rng.integers(-2, 3, size=(B, N, 3)).astype(np.float32) * 0.25
    tgt[:, ::3] += 0.125                                                              # targets between sites: four / eight equidistant sites
    wi, ww = oracle.three_nn(tgt, lattice)
    idx, w = ops.three_nn(T(tgt, dev), T(lattice, dev))
    assert np.array_equal(idx.cpu().numpy(), wi)
    assert np.array_equal(w.cpu().numpy(), ww)


# ------------------------------------------------------------------------------------ fixtures at full size
def test_full_size_digests(dev):
    big = json.load(open(os.path.join(GOLD, "full_size_digests.json")))
    for cfg in ("cfg2_N1024", "cfg2_N4096", "cfg4_N2048"):
        c = big[cfg]
        pts = synth.gauss_ball(c["B"], c["N"], c["seed"])
        assert sha(pts) == c["xyz_sha"]
        i1, x1 = ops.furthest_point_sample(T(pts, dev), 512, tie_stride=c["tie_stride"])
        i2, x2 = ops.furthest_point_sample(x1, 128, tie_stride=c["tie_stride"])
        b1 = ops.ball_query(x1, T(pts, dev), 0.2, 64)
        b2 = ops.ball_query(x2, x1, 0.4, 64)
        assert sha(i1.cpu().numpy()) == c["fps1_sha"], cfg
        assert sha(i2.cpu().numpy()) == c["fps2_sha"], cfg
        assert sha(b1.cpu().numpy()) == c["bq1_sha"], cfg
        assert sha(b2.cpu().numpy()) == c["bq2_sha"], cfg
    c = big["cfg3_knn_xyz"]
    pts = synth.gauss_ball(c["B"], c["N"], c["seed"])
    x = T(np.ascontiguousarray(pts.transpose(0, 2, 1)), dev)
    assert sha(ops.knn_indices(x, x, c["k"]).cpu().numpy()) == c["knn_sha"]


def test_degenerate_sizes(oracle, dev):
    """Smallest legal shapes of every index op: one point, one sample, one neighbour, one cloud."""
    one = np.array([[[0.5, -0.25, 0.125]]], np.float32)                       # B=1, N=1
    idx, new_xyz = ops.furthest_point_sample(T(one, dev), 1)
    assert idx.cpu().numpy().tolist() == [[0]] and np.array_equal(new_xyz.cpu().numpy(), one)
    bq, cnt = ops.ball_query(T(one, dev), T(one, dev), 0.1, 1, return_cnt=True)
    assert bq.cpu().numpy().tolist() == [[[0]]] and cnt.cpu().numpy().tolist() == [[1]]
    assert ops.knn_indices(T(one.transpose(0, 2, 1), dev), T(one.transpose(0, 2, 1), dev), 1).cpu().numpy().tolist() == [[[0]]]
    g = ops.group_points(T(one, dev), T(one, dev), T(one, dev), bq, True).cpu().numpy()
    assert g.shape == (1, 1, 1, 6) and np.array_equal(g[0, 0, 0], np.concatenate([np.zeros(3, np.float32), one[0, 0]]))
    pts = synth.gauss_ball(1, 3, 5)                                            # m == N == 3: a permutation of the cloud
    for S in (1, 2):
        want = oracle.fps(pts, 3, block_size=S)
        assert np.array_equal(ops.furthest_point_sample(T(pts, dev), 3, tie_stride=S)[0].cpu().numpy(), want)
    idx = ops.ball_query(T(pts, dev), T(pts, dev), 10.0, 5).cpu().numpy()     # ns > N: every row = 0,1,2 padded with 0
    assert np.array_equal(idx, oracle.ball_query(pts, pts, 10.0, 5))
    kk = ops.knn_indices(T(pts.transpose(0, 2, 1), dev), T(pts.transpose(0, 2, 1), dev), 3).cpu().numpy()
    assert np.array_equal(kk, oracle.knn(np.ascontiguousarray(pts.transpose(0, 2, 1)), np.ascontiguousarray(pts.transpose(0, 2, 1)), 3))
    with pytest.raises((ValueError, RuntimeError)):
        ops.furthest_point_sample(T(pts, dev), 4)                              # n_samples > N  (misc/ops.py:269)


def test_sa_level_matches_golden_fixture(dev):
    """One set-abstraction level through the product path (FPS -> ball query -> duplicate-compacted grouped rows -> fused MLP
    -> max) against the committed fixture: indices exact, pooled features within 1e-5 of the fp64 values."""
    from pointcloudlib_amd.networks.cls.pointnet2 import PointnetModule
    g = np.load(os.path.join(GOLD, "sa_level.npz"))
    spec = [int(g["feat"].shape[-1])] + [int(g[f"w{i}"].shape[0]) for i in range(3)]
    mod = PointnetModule(spec, n_points=int(g["n_points"]), radius=float(g["radius"]), n_samples=int(g["n_samples"])).to(dev).train()
    mlp = mod.mlps[0]
    with torch.no_grad():
        for i in range(3):
            assert mlp.weights[i].shape == g[f"w{i}"].shape
            mlp.weights[i].copy_(torch.from_numpy(g[f"w{i}"])); mlp.gammas[i].copy_(torch.from_numpy(g[f"gamma{i}"]))
            mlp.betas[i].copy_(torch.from_numpy(g[f"beta{i}"]))
    mod.sampler.tie_stride = int(g["tie_stride"])
    xyz, feat = T(g["xyz"], dev), T(g["feat"], dev)
    new_xyz, idxs = mod.sample(xyz)
    assert np.array_equal(new_xyz.cpu().numpy(), g["new_xyz"])
    assert np.array_equal(idxs[0][0].cpu().numpy(), g["bq_idx"])
    grouped = ops.group_points(xyz, new_xyz, feat, idxs[0][0], True)
    assert np.array_equal(grouped.cpu().numpy(), g["grouped"])
    for compact in (True, False):
        mod.compact_duplicates = compact
        _, pooled = mod(xyz, feat)
        err = np.abs(pooled.detach().double().cpu().numpy() - g["pooled_f64"]).max()
        assert err <= 1e-5 * max(1.0, np.abs(g["pooled_f64"]).max()), (compact, err)


@pytest.mark.parametrize("B,N,S,k", [(2, 512, 128, 64), (3, 1024, 512, 32), (2, 1024, 40, 100), (1, 2048, 33, 16), (2, 64, 64, 64), (1, 4096, 20, 8)])
def test_knn_point_matmul_form_matches_oracle(oracle, dev, B, N, S, k):
    """pcl_knn_point_matmul_f32 (PointConv's knn_point in the reference's own -2ab + a^2 + b^2 arithmetic, misc/pointconv_utils.py:34-53,
    :120-131) against oracle.knn_point_matmul: exact [B,S,k] lists for both dot-product readings, incl. duplicate points (ties by
    index), k > 64 (extraction rounds) and k = N."""
    from pointcloudlib_amd import _lib
    from pointcloudlib_amd.misc.ops import _p, _stream
    pts = synth.gauss_ball(B, N, 300 + N + k)
    pts[:, 5] = pts[:, 17]                                  # exact duplicates: distance ties, lower index first
    pts[:, N // 2: N // 2 + 3] = pts[:, 1:2]
    q = np.ascontiguousarray(pts[:, np.random.default_rng(k).permutation(N)[:S]])
    x, qd = T(pts, dev), T(q, dev)
    for fma_dot in (1, 0):
        want = oracle.knn_point_matmul(k, pts, q, fma_dot=bool(fma_dot))
        out = torch.empty((B, S, k), dtype=torch.int32, device=dev)
        _lib.call("pcl_knn_point_matmul_f32", _p(x), _p(qd), B, N, S, k, fma_dot, _p(out), _stream())
        assert np.array_equal(out.cpu().numpy(), want), (B, N, S, k, fma_dot)


@pytest.mark.parametrize("B,N,m,radii,ns", [(4, 2048, 512, [0.1, 0.2, 0.4], [16, 32, 128]), (3, 512, 128, [0.2, 0.4, 0.8], [32, 64, 128]),
                                            (2, 1000, 77, [0.05, 10.0], [8, 4]), (2, 300, 300, [0.3], [16]),
                                            (1, 64, 5, [1e-6, 0.2, 0.25, 0.3], [4, 1, 70, 64])])
def test_ball_query_multi_equals_one_query_per_radius(oracle, dev, B, N, m, radii, ns):
    """pcl_ball_query_multi_f32 (multi-scale grouping: every scale's list out of one scan of the cloud) against pcl_ball_query_f32 per
    radius and against the oracle: lists and counts identical, including radii with no hit at all, radii that fill their list in the
    first 64 points (early exit per radius) and sample counts above / below the wave width."""
    xyz = synth.gauss_ball(B, N, 11 + N + m)
    cent = synth.gauss_ball(B, m, 5 + m)
    h = min(m // 2, N)
    cent[:, :h] = xyz[:, :h]                                   # half of the centres are cloud points (as after FPS), half are not
    X, Q = T(xyz, dev), T(cent, dev)
    got = ops.ball_query_multi(Q, X, radii, ns, return_cnt=True)
    assert len(got) == len(radii)
    for (idx, cnt), r, s in zip(got, radii, ns):
        one_idx, one_cnt = ops.ball_query(Q, X, r, s, return_cnt=True)
        assert torch.equal(idx, one_idx) and torch.equal(cnt, one_cnt), r
        np.testing.assert_array_equal(idx.cpu().numpy(), oracle.ball_query(cent, xyz, r, s))
    offs = ops.group_offsets_multi([c for _, c in got])                       # the scales' row offsets in one launch
    for off, (_, c) in zip(offs, got):
        assert torch.equal(off, ops.group_offsets(c))
        want = np.concatenate([[0], np.cumsum(np.maximum(c.cpu().numpy().reshape(-1), 1))]).astype(np.int32)
        np.testing.assert_array_equal(off.cpu().numpy(), want)
    only = ops.ball_query_multi(Q, X, radii, ns)
    for a, (b, _) in zip(only, got):
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        ops.ball_query_multi(Q, X, [0.1] * 5, [4] * 5)


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,N,k", [(4, 3, 1024, 20), (2, 64, 1000, 20), (3, 128, 513, 40), (1, 7, 64, 64), (2, 16, 2048, 5)])
def test_knn_lists_written_as_rows_equal_the_permuted_search(B, C, N, k):
    """pcl_knn_nk_f32: the fused search writing [B, Nq, k] rows (round 6) is the reference-layout search permuted -- index for index."""
    import torch
    from pointcloudlib_amd.misc import ops
    torch.manual_seed(31)
    x = torch.randn(B, C, N, device="cuda")
    x[:, :, 5] = x[:, :, 3]                                    # an exact tie per query
    a = ops.knn_indices(x, x, k).permute(0, 2, 1).contiguous()
    b = ops.knn_lists(x, x, k)
    assert b.shape == (B, N, k) and b.dtype == torch.int32 and torch.equal(a, b)
