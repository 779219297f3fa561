"""Synthetic point clouds with the statistics of the reference's ModelNet40 pipeline (no dataset on the box).

``gauss_ball`` mirrors data_utils/modelnet40_loader.py: centre + scale into the unit sphere (:121-125), then
the train-time anisotropic scale U[2/3,3/2]^3 and shift U[-0.2,0.2]^3 (:128-132).  ``sphere_shell`` is the
never-saturating worst case for ball query.  NumPy only; seeds are explicit.
"""
import numpy as np


def gauss_ball(B, N, seed, augment=True):
    rng = np.random.default_rng(seed)
    p = rng.standard_normal((B, N, 3)).astype(np.float32)
    p -= p.mean(axis=1, keepdims=True)
    p /= np.sqrt((p ** 2).sum(-1)).max(axis=1)[:, None, None]
    if augment:
        p = p * rng.uniform(2.0 / 3.0, 1.5, (B, 1, 3)).astype(np.float32) + rng.uniform(-0.2, 0.2, (B, 1, 3)).astype(np.float32)
    return np.ascontiguousarray(p, dtype=np.float32)


def sphere_shell(B, N, seed):
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((B, N, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    r = rng.uniform(0.5, 1.0, (B, N, 1))
    return np.ascontiguousarray(d * r, dtype=np.float32)


def unit_normals(B, N, seed):
    rng = np.random.default_rng(seed)
    n = rng.standard_normal((B, N, 3))
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    return np.ascontiguousarray(n, dtype=np.float32)


def labels(B, n_classes, seed):
    return np.random.default_rng(seed).integers(0, n_classes, B).astype(np.int64)


def adversarial_clouds(seed=0):
    """Small correctness-only clouds: exact duplicates, lattice ties, near-origin points, dense clusters,
    isolated points."""
    rng = np.random.default_rng(seed)
    out = {}
    lat = np.stack(np.meshgrid(*[np.arange(4)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32) * 0.25 - 0.375
    out["lattice"] = np.stack([lat, lat[::-1].copy(), rng.permutation(lat)], 0)
    dup = rng.standard_normal((3, 24, 3)).astype(np.float32) * 0.4
    out["duplicates"] = np.concatenate([dup, dup[:, ::-1], dup[:, :16]], axis=1)
    near = rng.standard_normal((3, 64, 3)).astype(np.float32) * 0.5
    near[:, ::5] *= 0.01                                    # squared norm << 1e-3 -> skipped by FPS
    near[:, 0] = 0.0
    out["near_origin"] = near
    out["all_skipped"] = (rng.standard_normal((2, 32, 3)) * 0.005).astype(np.float32)
    clus = rng.standard_normal((3, 96, 3)).astype(np.float32) * 0.02
    clus[:, 80:] += rng.standard_normal((3, 16, 3)).astype(np.float32) * 2.0   # isolated outliers
    clus += 0.3
    out["cluster_outliers"] = clus
    return out
