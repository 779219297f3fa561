"""Second, independent restatement of the hot path in NumPy -- closed forms, no thread emulation.

Test infrastructure only.  Written from the *derived rules*  rather than from the
kernel text, so that a common-mode misreading in ``pcl_oracle.c`` (which emulates the CUDA threads
literally) shows up as a disagreement:

* FPS (misc/ops.py:124-234): winner of each step = arg-max of the running min-distance over the
  non-skipped points, ties resolved by the smallest ``(bitreverse_{log2 S}(k mod S), k)`` where S is the
  reference's launch block size; all-skipped -> index 0.
* ball query (misc/ops.py:291-330): first ``nsample`` indices (ascending) with d2 < fl(r*r), padded
  with the first hit; no hit -> zeros.
* KNN (misc/ops.py:429-552): k smallest by (distance, index), ascending.
All arithmetic is done with float32 arrays so every operation is single-rounded as in the C source.
"""
import numpy as np

F = np.float32


def _bitrev(v, bits):
    r = np.zeros_like(v)
    for i in range(bits):
        r |= ((v >> i) & 1) << (bits - 1 - i)
    return r


def tie_rank(N, S):
    """Priority of index k among exactly tied candidates: smaller rank wins."""
    k = np.arange(N, dtype=np.int64)
    bits = int(S).bit_length() - 1
    assert (1 << bits) == S
    return _bitrev(k % S, bits) * N + k  # (bitrev(k mod S), k) lexicographic


def fps_np(xyz, m, S=1, skip=True, start_idx=None):
    xyz = np.asarray(xyz, F)
    B, N, _ = xyz.shape
    out = np.zeros((B, m), np.int32)
    rank = tie_rank(N, S)
    for b in range(B):
        p = xyz[b]
        x, y, z = p[:, 0], p[:, 1], p[:, 2]
        if skip:
            mag = (x * x + y * y) + z * z                      # float32, left-to-right
            live = ~(mag.astype(np.float64) <= 1e-3)
        else:
            live = np.ones(N, bool)
        temp = np.full(N, F(1e10), F)
        old = 0 if start_idx is None else int(start_idx[b])
        out[b, 0] = old
        for j in range(1, m):
            dx, dy, dz = x - x[old], y - y[old], z - z[old]
            d = (dx * dx + dy * dy) + dz * dz
            temp = np.where(live, np.minimum(d, temp), temp)
            if not live.any():
                old = 0
            else:
                cand = np.where(live, temp, F(-1))
                best = cand.max()
                tied = np.flatnonzero(cand == best)
                old = int(tied[np.argmin(rank[tied])])
            out[b, j] = old
    return out


def ball_query_np(new_xyz, xyz, radius, nsample):
    new_xyz = np.asarray(new_xyz, F)
    xyz = np.asarray(xyz, F)
    B, m, _ = new_xyz.shape
    r = F(radius)
    r2 = F(r * r)
    idx = np.zeros((B, m, nsample), np.int32)
    cnt = np.zeros((B, m), np.int32)
    for b in range(B):
        q = new_xyz[b][:, None, :]
        p = xyz[b][None, :, :]
        dx = q[..., 0] - p[..., 0]
        dy = q[..., 1] - p[..., 1]
        dz = q[..., 2] - p[..., 2]
        d2 = (dx * dx + dy * dy) + dz * dz
        hit = d2 < r2
        for j in range(m):
            h = np.flatnonzero(hit[j])[:nsample]
            cnt[b, j] = len(h)
            if len(h):
                idx[b, j, :] = h[0]
                idx[b, j, :len(h)] = h
    return idx, cnt


def group_np(xyz, new_xyz, feat, idx, use_xyz=True):
    B = idx.shape[0]
    bi = np.arange(B)[:, None, None]
    parts = []
    if use_xyz:
        parts.append(np.asarray(xyz, F)[bi, idx] - np.asarray(new_xyz, F)[:, :, None, :])
    if feat is not None:
        parts.append(np.asarray(feat, F)[bi, idx])
    return np.concatenate(parts, -1)


def knn_np(x_q, x_r, k):
    x_q = np.asarray(x_q, F)
    x_r = np.asarray(x_r, F)
    B, C, Nq = x_q.shape
    Nr = x_r.shape[2]
    out = np.zeros((B, k, Nq), np.int32)
    for b in range(B):
        ssd = np.zeros((Nr, Nq), F)
        for c in range(C):
            tmp = x_r[b, c][:, None] - x_q[b, c][None, :]
            ssd = ssd + tmp * tmp
        order = np.argsort(ssd, axis=0, kind="stable")          # stable -> ties by lower r
        out[b] = order[:k].astype(np.int32)
    return out


def three_nn_np(xyz1, xyz2):
    xyz1 = np.asarray(xyz1, F)
    xyz2 = np.asarray(xyz2, F)
    a = xyz1[:, :, None, :]
    b = xyz2[:, None, :, :]
    dx, dy, dz = a[..., 0] - b[..., 0], a[..., 1] - b[..., 1], a[..., 2] - b[..., 2]
    d = (dx * dx + dy * dy) + dz * dz
    order = np.argsort(d, axis=-1, kind="stable")[..., :3]
    dd = np.take_along_axis(d, order, -1)
    rec = F(1.0) / (dd + F(1e-8))
    norm = (rec[..., 0] + rec[..., 1]) + rec[..., 2]
    return order.astype(np.int32), rec / norm[..., None]
