"""ShapeNet-Part input pipeline -- counterpart of /root/reference/data_utils/shapenet_loader.py.

Items (:87-97): the first ``num_points`` points of a shape, its category id and per-point part ids; the 'trainval'
partition shuffles the points of every item (the upstream translation augmentation is commented out there and is not
applied here either).  Batches are ``(points [B,n,3] f32, label [B,1] i64, seg [B,n] i64)``, ``drop_last=True`` (:80-85).
``class_choice`` restricts to one category and re-bases the part ids to ``seg_start_index`` exactly as upstream (:66-76).

Files: the dataset's ``shapenet_part_seg_hdf5_data/*{train,val,test}*.h5`` with keys ``data``, ``label``, ``pid``
(:27-49).  h5py is not part of this image, so the reader takes h5py when it imports and otherwise the same arrays saved
as ``.npz`` next to (or instead of) each ``.h5`` (``np.savez(name.npz, data=..., label=..., pid=...)``); nothing is
downloaded (no network).  ``SyntheticShapeNetPart`` yields ``synth.gauss_ball`` clouds with random labels through the
same item recipe for throughput runs.
"""
import glob
import os

import numpy as np
import torch

from .modelnet40_loader import _Batches

CAT2ID = {"airplane": 0, "bag": 1, "cap": 2, "car": 3, "chair": 4, "earphone": 5, "guitar": 6, "knife": 7, "lamp": 8,
          "laptop": 9, "motor": 10, "mug": 11, "pistol": 12, "rocket": 13, "skateboard": 14, "table": 15}
SEG_NUM = [4, 2, 2, 4, 4, 3, 3, 2, 4, 2, 6, 2, 3, 3, 3, 3]
INDEX_START = [0, 4, 6, 8, 12, 16, 19, 22, 24, 28, 30, 36, 38, 41, 44, 47]


def _read_arrays(path):
    if path.endswith(".npz"):
        with np.load(path) as z:
            return z["data"], z["label"], z["pid"]
    try:
        import h5py
    except ImportError as e:
        raise ImportError(f"{path}: h5py is not installed; convert the file once to .npz (keys data, label, pid)") from e
    with h5py.File(path, "r") as f:
        return f["data"][:], f["label"][:], f["pid"][:]


def load_data_partseg(partition, root=None):
    root = root or os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
    parts = ["train", "val"] if partition == "trainval" else [partition]
    files = []
    for p in parts:
        found = sorted(glob.glob(os.path.join(root, "shapenet*hdf5*", f"*{p}*.h5")) +
                       glob.glob(os.path.join(root, "shapenet*hdf5*", f"*{p}*.npz")))
        stems = set()
        for f in found:                                   # a shard present in both formats is read once
            stem = os.path.splitext(f)[0]
            if stem not in stems:
                stems.add(stem)
                files.append(f)
    if not files:
        raise FileNotFoundError(f"{root}: no shapenet_part_seg_hdf5_data/*{partition}* shards here (and no network to fetch them); "
                                "use SyntheticShapeNetPart for throughput runs")
    data, label, seg = zip(*(_read_arrays(f) for f in files))
    return (np.concatenate(data).astype(np.float32), np.concatenate(label).astype(np.int64).reshape(-1, 1),
            np.concatenate(seg).astype(np.int64))


class ShapeNetPart:
    def __init__(self, num_points, partition="train", class_choice=None, batch_size=16, shuffle=False, root=None):
        self.num_points, self.partition, self.class_choice = num_points, partition, class_choice
        self.batch_size, self.shuffle = batch_size, shuffle
        self.cat2id, self.seg_num, self.index_start = CAT2ID, SEG_NUM, INDEX_START
        self.data, self.label, self.seg = self._load(root)
        if class_choice is not None:
            cid = self.cat2id[class_choice]
            keep = (self.label == cid).reshape(-1)
            self.data, self.label, self.seg = self.data[keep], self.label[keep], self.seg[keep]
            self.seg_num_all, self.seg_start_index = self.seg_num[cid], self.index_start[cid]
        else:
            self.seg_num_all, self.seg_start_index = 50, 0
        self._batches = _Batches(self.data.shape[0], batch_size, shuffle, drop_last=True)

    def _load(self, root):
        return load_data_partseg(self.partition, root)

    def __len__(self):
        return len(self._batches)

    def item(self, i):
        pts, seg = self.data[i][:self.num_points], self.seg[i][:self.num_points]
        if self.partition == "trainval":
            order = np.arange(pts.shape[0])
            np.random.shuffle(order)
            pts, seg = pts[order], seg[order]
        return pts, self.label[i], seg

    def __getitem__(self, i):
        return self.item(i)

    def __iter__(self):
        for ids in self._batches.order():
            items = [self.item(int(i)) for i in ids]
            yield (torch.from_numpy(np.stack([b[0] for b in items])), torch.from_numpy(np.stack([b[1] for b in items])),
                   torch.from_numpy(np.stack([b[2] for b in items])))


class SyntheticShapeNetPart(ShapeNetPart):
    """Same recipe on synthetic shapes: ``synth.gauss_ball`` clouds, a random category per shape and part ids drawn from
    that category's id range (so ``calculate_shape_IoU`` sees valid parts)."""

    def __init__(self, num_points, partition="train", class_choice=None, batch_size=16, shuffle=False, n_items=64, seed=0):
        self._n_items, self._seed, self._n_raw = n_items, seed, max(num_points, 2048)
        super().__init__(num_points, partition, class_choice, batch_size, shuffle)

    def _load(self, root):
        from .. import synth
        rng = np.random.default_rng(self._seed)
        data = synth.gauss_ball(self._n_items, self._n_raw, self._seed + 77)
        label = rng.integers(0, 16, (self._n_items, 1)).astype(np.int64)
        lo = np.asarray(INDEX_START)[label]                                       # [n,1]
        seg = lo + rng.integers(0, 1 << 30, (self._n_items, self._n_raw)) % np.asarray(SEG_NUM)[label]
        return data.astype(np.float32), label, seg.astype(np.int64)
