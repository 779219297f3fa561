"""ModelNet40 (normal-resampled) input pipeline -- counterpart of /root/reference/data_utils/modelnet40_loader.py.

Same per-item recipe (:86-103): shuffle the first ``n_points`` points, split xyz / normals, centre + scale to the unit
sphere (``normalize_pointclouds`` :121-125), and for training an anisotropic scale U[2/3,3/2]^3 and shift U[-0.2,0.2]^3
(``translate_pointcloud`` :128-132); ``random_point_dropout`` (:105-113) exists upstream but is commented out of the
pipeline and is kept here as a function only.  Batches are ``(pts [B,n,3], normals [B,n,3], cls [B])`` -- note that the
reference's ``collect_batch`` (:115-119) drops the normals; the training script gets them from the item tuple instead.

Reads the dataset's text files directly (``<root>/modelnet40_normal_resampled/<shape>/<id>.txt``, comma separated
x,y,z,nx,ny,nz); the reference's lmdb/msgpack cache and its download step are not reproduced (no network here).
``SyntheticModelNet40`` yields ``synth.gauss_ball`` clouds through the same recipe when no dataset is present.
"""
import os

import numpy as np
import torch


def normalize_pointclouds(pts):
    pts = pts - pts.mean(axis=0)
    scale = np.sqrt((pts ** 2).sum(axis=1).max())
    return pts / scale


def translate_pointcloud(pointcloud, rng=np.random):
    xyz1 = rng.uniform(low=2. / 3., high=3. / 2., size=[3])
    xyz2 = rng.uniform(low=-0.2, high=0.2, size=[3])
    return np.add(np.multiply(pointcloud, xyz1), xyz2).astype("float32")


def random_point_dropout(pc, normal, max_dropout_ratio=0.875, rng=np.random):
    """Upstream (disabled) augmentation: dropped points become copies of point 0 -- exact duplicates, which is what the
    FPS tie rule and the duplicate-compacted grouping are tested against."""
    dropout_ratio = rng.random() * max_dropout_ratio
    drop_idx = np.where(rng.random((pc.shape[0])) <= dropout_ratio)[0]
    if len(drop_idx) > 0:
        pc[drop_idx, :] = pc[0, :]
        normal[drop_idx, :] = normal[0, :]
    return pc, normal


class _Batches:
    def __init__(self, n_items, batch_size, shuffle, drop_last=False):
        self.n_items, self.batch_size, self.shuffle, self.drop_last = n_items, batch_size, shuffle, drop_last

    def __len__(self):
        return self.n_items // self.batch_size if self.drop_last else -(-self.n_items // self.batch_size)

    def order(self):
        idx = np.arange(self.n_items)
        if self.shuffle:
            np.random.shuffle(idx)
        return [idx[i:i + self.batch_size] for i in range(0, self.n_items, self.batch_size)
                if not self.drop_last or i + self.batch_size <= self.n_items]


class ModelNet40:
    def __init__(self, n_points, train, batch_size=1, shuffle=False, root=None, cache_dir=None):
        """``cache_dir``: where to keep ``.npy`` copies of the parsed text files (None: in memory only; nothing is ever written
        into the dataset directory)."""
        self.n_points, self.train, self.batch_size, self.shuffle = n_points, train, batch_size, shuffle
        self.cache_dir = cache_dir
        root = root or os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
        self.path = os.path.join(root, "modelnet40_normal_resampled")
        names = os.path.join(self.path, "modelnet40_shape_names.txt")
        if not os.path.exists(names):
            raise FileNotFoundError(f"{self.path}: ModelNet40 (normal resampled) is not here and cannot be downloaded; "
                                    "use SyntheticModelNet40 for throughput runs")
        cats = [line.rstrip() for line in open(names)]
        self.classes = dict(zip(cats, range(len(cats))))
        split = "train" if train else "test"
        self.shapes = []
        for line in open(os.path.join(self.path, f"modelnet40_{split}.txt")):
            shape_id = line.rstrip()
            self.shapes.append(("_".join(shape_id.split("_")[:-1]), shape_id + ".txt"))
        self._batches = _Batches(len(self.shapes), batch_size, shuffle)

    def __len__(self):
        return len(self._batches)

    def load(self, idx):
        """One shape, parsed ONCE: the 10 000-line text file costs ~0.1 s in np.loadtxt, against a ~13k clouds/s training
        step.  The rows an item uses (the first ``n_points``) are kept in memory; with ``cache_dir`` the parsed array is also
        kept there as ``<shape>.npy`` (the reference caches the parsed dataset in LMDB/msgpack,
        data_utils/modelnet40_loader.py:40-100)."""
        cache = self.__dict__.setdefault("_cache", {})
        hit = cache.get(idx)
        if hit is not None:
            return hit
        shape_name, shape_file = self.shapes[idx]
        txt = os.path.join(self.path, shape_name, shape_file)
        # on-disk copy of the parsed array: only when a cache directory was asked for (``cache_dir``), never inside the dataset
        cdir = getattr(self, "cache_dir", None)
        npy = os.path.join(cdir, os.path.splitext(shape_file)[0] + ".npy") if cdir else None
        pts = None
        if npy and os.path.exists(npy) and os.path.getmtime(npy) >= os.path.getmtime(txt):
            try:
                pts = np.load(npy)
            except (OSError, ValueError):
                pts = None
        if pts is None:
            pts = np.loadtxt(txt, delimiter=",", dtype=np.float32)
        if pts is not None and npy and not os.path.exists(npy):
            try:
                os.makedirs(cdir, exist_ok=True)
                tmp = npy + f".{os.getpid()}.tmp"
                with open(tmp, "wb") as fh:
                    np.save(fh, pts)
                os.replace(tmp, npy)                      # atomic: concurrent loaders never see a partial file
            except OSError:
                pass
        # item() uses the first n_points rows only: keep those (24 KB per shape at 1024 points, not the 240 KB of all 10 000 rows)
        cache[idx] = (np.ascontiguousarray(pts[:self.n_points]), self.classes[shape_name])
        return cache[idx]

    def item(self, idx):
        pts, cls = self.load(idx)
        pt_idxs = np.arange(0, self.n_points)
        np.random.shuffle(pt_idxs)
        pts = pts[pt_idxs, :]
        pts, normals = pts[:, :3], pts[:, 3:]
        pts = normalize_pointclouds(pts)
        if self.train:
            pts = translate_pointcloud(pts)
        return pts.astype(np.float32), normals.astype(np.float32), cls

    def __iter__(self):
        for ids in self._batches.order():
            items = [self.item(int(i)) for i in ids]
            yield (torch.from_numpy(np.stack([b[0] for b in items])), torch.from_numpy(np.stack([b[1] for b in items])),
                   torch.from_numpy(np.array([b[2] for b in items], dtype=np.int64)))


class SyntheticModelNet40(ModelNet40):
    """Same item recipe on synthetic clouds (``synth.gauss_ball`` + random unit normals, random labels)."""

    def __init__(self, n_points, train, batch_size=1, shuffle=False, n_items=256, n_classes=40, seed=0):  # noqa: D107
        self.n_points, self.train, self.batch_size, self.shuffle = n_points, train, batch_size, shuffle
        rng = np.random.default_rng(seed)
        raw = rng.standard_normal((n_items, max(n_points, 1024), 6)).astype(np.float32)
        raw[..., 3:] /= np.linalg.norm(raw[..., 3:], axis=-1, keepdims=True)
        self._raw = raw
        self._labels = rng.integers(0, n_classes, n_items)
        self.classes = {str(i): i for i in range(n_classes)}
        self.shapes = [(str(int(c)), "") for c in self._labels]
        self._batches = _Batches(n_items, batch_size, shuffle)

    def load(self, idx):
        return self._raw[idx], int(self._labels[idx])
