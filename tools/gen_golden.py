"""Generate tests/golden/*.npz -- regression fixtures for the hot path.

The reference holds no golden vectors and cannot be imported here (Jittor is absent; its kernels are CUDA
text), so these are produced by the repo's own oracle (oracle/pcl_oracle.c), after it agreed with the
independent NumPy restatement (oracle/np_oracle.py) on every case.  They are data only: seeded inputs and
the expected indices / grouped tensors; large index arrays are stored as SHA-256 digests.

    python tools/gen_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from oracle import np_oracle as npo  # noqa: E402
from pointcloudlib_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    os.makedirs(OUT, exist_ok=True)
    small = {}
    # --- adversarial small clouds: FPS with every tie stride, ball query, group
    for name, pts in synth.adversarial_clouds(0).items():
        B, N, _ = pts.shape
        m = max(2, N // 3)
        small[f"{name}.xyz"] = pts
        for S in (1, 2, 4, 8, 16):
            a = oracle.fps(pts, m, block_size=S)
            assert np.array_equal(a, npo.fps_np(pts, m, S)), (name, S)
            small[f"{name}.fps_S{S}"] = a
        idx, new_xyz = oracle.fps(pts, m, block_size=1, return_xyz=True)
        for r, ns in ((0.1, 4), (0.3, 8), (1.0, 16)):
            bq, cnt = oracle.ball_query(new_xyz, pts, r, ns, return_cnt=True)
            b2, c2 = npo.ball_query_np(new_xyz, pts, r, ns)
            assert np.array_equal(bq, b2) and np.array_equal(cnt, c2)
            small[f"{name}.bq_r{r}_ns{ns}"] = bq
            small[f"{name}.bqcnt_r{r}_ns{ns}"] = cnt
    # --- knn small
    rng = np.random.default_rng(7)
    q = rng.integers(-2, 3, (2, 5, 20)).astype(np.float32)
    r = rng.integers(-2, 3, (2, 5, 33)).astype(np.float32)
    small["knn.q"], small["knn.r"] = q, r
    for k in (1, 7, 33):
        a = oracle.knn(q, r, k)
        assert np.array_equal(a, npo.knn_np(q, r, k))
        small[f"knn.k{k}"] = a
    np.savez_compressed(os.path.join(OUT, "small_cases.npz"), **small)

    # --- full-size synthetic batches (inputs regenerated from the seed; outputs as digests)
    big = {}
    for cfg, (B, N) in {"cfg2_N1024": (32, 1024), "cfg2_N4096": (32, 4096), "cfg4_N2048": (16, 2048)}.items():
        pts = synth.gauss_ball(B, N, 20242)
        S = oracle.optimal_block(B)
        i1, x1 = oracle.fps(pts, 512, block_size=S, return_xyz=True)
        i2, x2 = oracle.fps(x1, 128, block_size=S, return_xyz=True)
        b1 = oracle.ball_query(x1, pts, 0.2, 64)
        b2 = oracle.ball_query(x2, x1, 0.4, 64)
        big[cfg] = {"B": B, "N": N, "seed": 20242, "tie_stride": S, "xyz_sha": sha(pts),
                    "fps1_sha": sha(i1), "fps2_sha": sha(i2), "bq1_sha": sha(b1), "bq2_sha": sha(b2),
                    "fps1_head": i1[0, :8].tolist(), "bq1_row0": b1[0, 0, :8].tolist()}
    # DGCNN-style knn on xyz, k=20
    pts = synth.gauss_ball(8, 1024, 20243)
    x = np.ascontiguousarray(pts.transpose(0, 2, 1))
    kk = oracle.knn(x, x, 20)
    big["cfg3_knn_xyz"] = {"B": 8, "N": 1024, "seed": 20243, "k": 20, "knn_sha": sha(kk), "head": kk[0, :4, 0].tolist()}
    json.dump(big, open(os.path.join(OUT, "full_size_digests.json"), "w"), indent=1)

    # --- one set-abstraction level end to end: (input, weights) -> (idx, grouped, pooled); SURVEY section 8a row 8.
    # Restatement of PointnetModule.execute (networks/cls/pointnet2.py:45-57) in oracle/cpu_model.py, evaluated in fp64
    # (the yardstick of the 1e-5 feature tolerance) and in fp32.
    import torch
    from oracle.cpu_model import sa_module_cpu
    g = torch.Generator().manual_seed(20245)
    B, N, m, ns, radius, S = 3, 256, 48, 16, 0.35, 2
    xyz = synth.gauss_ball(B, N, 20245)
    feat = torch.randn(B, N, 5, generator=g).numpy().astype(np.float32)
    spec = [8, 32, 32, 64]
    ws = [(torch.randn(spec[i + 1], spec[i], generator=g) / spec[i] ** 0.5).numpy().astype(np.float32) for i in range(3)]
    gs = [torch.empty(c).uniform_(0.5, 1.5, generator=g).numpy().astype(np.float32) for c in spec[1:]]
    bs = [torch.empty(c).uniform_(-0.3, 0.3, generator=g).numpy().astype(np.float32) for c in spec[1:]]
    gs[1][::3] *= -1.0                               # negative gamma: the max must not assume a monotone BatchNorm
    def level(dt):
        T = lambda a: torch.from_numpy(a).to(dt)
        return sa_module_cpu(T(xyz), T(feat), [T(w) for w in ws], [T(x) for x in gs], [T(x) for x in bs], m, radius, ns, S, return_aux=True)
    nx64, y64, aux = level(torch.float64)
    nx32, y32, _ = level(torch.float32)
    sa = {"xyz": xyz, "feat": feat, "n_points": m, "radius": radius, "n_samples": ns, "tie_stride": S,
          "fps_idx": aux["fps_idx"], "bq_idx": aux["bq_idx"], "new_xyz": nx32.numpy(), "grouped": aux["grouped"].float().numpy(),
          "pooled_f64": y64.numpy(), "pooled_f32": y32.numpy()}
    for i in range(3):
        sa[f"w{i}"], sa[f"gamma{i}"], sa[f"beta{i}"] = ws[i], gs[i], bs[i]
    np.savez_compressed(os.path.join(OUT, "sa_level.npz"), **sa)
    print("wrote", os.listdir(OUT))


if __name__ == "__main__":
    main()
