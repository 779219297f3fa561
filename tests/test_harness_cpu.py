"""CPU: the training-harness pieces of SURVEY section 8(f) rank 4 -- input recipe of the ModelNet40 loader, shape IoU."""
import numpy as np


def test_loader_recipe_matches_reference_arithmetic():
    from pointcloudlib_amd.data_utils.modelnet40_loader import (SyntheticModelNet40, normalize_pointclouds,
                                                                 random_point_dropout, translate_pointcloud)
    rng = np.random.default_rng(0)
    p = rng.standard_normal((500, 3)).astype(np.float32) * 3 + 5
    q = normalize_pointclouds(p)
    assert np.allclose(q.mean(0), 0, atol=1e-6) and abs(np.sqrt((q ** 2).sum(1)).max() - 1) < 1e-6      # :121-125
    t = translate_pointcloud(q)
    ratio = (t - t.mean(0)) / np.where(np.abs(q - q.mean(0)) > 1e-3, q - q.mean(0), np.nan)
    s = np.nanmedian(ratio, axis=0)
    assert ((s >= 2 / 3 - 1e-3) & (s <= 3 / 2 + 1e-3)).all() and t.dtype == np.float32                  # :128-132
    pc, nr = random_point_dropout(q.copy(), q.copy(), 0.875, np.random.default_rng(1))
    dropped = (pc == pc[0]).all(1)
    assert dropped.sum() >= 1 and (nr[dropped] == nr[0]).all()                                           # :105-113
    ds = SyntheticModelNet40(n_points=256, train=True, batch_size=8, shuffle=True, n_items=20)
    batches = list(ds)
    assert len(batches) == len(ds) == 3 and batches[0][0].shape == (8, 256, 3) and batches[-1][0].shape[0] == 4
    pts, normals, cls = batches[0]
    assert pts.dtype.is_floating_point and cls.dtype.__str__() == "torch.int64"
    assert np.allclose(np.linalg.norm(normals.numpy(), axis=-1), 1, atol=1e-5)


def test_shape_iou():
    from pointcloudlib_amd.train_utils import calculate_shape_IoU, index_start, seg_num
    seg = np.array([[0, 0, 1, 1, 2, 3], [4, 4, 4, 5, 5, 5]])
    pred = np.array([[0, 1, 1, 1, 2, 2], [4, 4, 4, 5, 5, 5]])
    label = np.array([[0], [1]])                       # airplane: parts 0-3, bag: parts 4-5
    ious = calculate_shape_IoU(pred, seg, label, None)
    # airplane: part0 1/2, part1 2/3, part2 1/2, part3 0/1
    assert abs(ious[0] - np.mean([1 / 2, 2 / 3, 1 / 2, 0.0])) < 1e-12 and ious[1] == 1.0
    assert calculate_shape_IoU(np.zeros((1, 4)), np.zeros((1, 4)), np.array([[2]]), None) == [1.0]   # empty unions count as 1
    assert len(seg_num) == len(index_start) == 16 and index_start[-1] + seg_num[-1] == 50


def test_affinity_helper_parses_topology(tmp_path, monkeypatch):
    """pin_to_gpu_node(): KFD topology -> NUMA node -> cpulist, on a fake sysfs; never raises on a host without one."""
    from pointcloudlib_amd import affinity
    assert affinity._cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    kfd = tmp_path / "kfd"
    for n, (simd, minor) in enumerate([(0, 0), (1024, 128), (1024, 129)]):
        (kfd / str(n)).mkdir(parents=True)
        (kfd / str(n) / "properties").write_text(f"cpu_cores_count 0\nsimd_count {simd}\ndrm_render_minor {minor}\n")
    monkeypatch.setattr(affinity, "_KFD", str(kfd))
    real_open = open

    def fake_open(path, *a, **k):
        if str(path).startswith("/sys/class/drm/renderD"):
            import io
            return io.StringIO("0\n" if "renderD128" in str(path) else "1\n")
        return real_open(path, *a, **k)
    monkeypatch.setattr("builtins.open", fake_open)
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert affinity.gpu_numa_nodes() == [0, 1]
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1")
    assert affinity.gpu_numa_nodes() == [1]
    monkeypatch.undo()
    import os
    before = os.sched_getaffinity(0)
    r = affinity.pin_to_gpu_node(5)              # no such GPU here: pins to the current node or does nothing
    assert r is None or "numa node" in r
    os.sched_setaffinity(0, before)


def test_shapenet_loader_recipe(tmp_path):
    """ShapeNetPart: npz shards with the h5 keys, trainval = train + val, drop_last batches, class_choice re-basing,
    per-item point shuffle only for 'trainval' (shapenet_loader.py:53-97)."""
    import numpy as np
    from pointcloudlib_amd.data_utils.shapenet_loader import INDEX_START, SEG_NUM, ShapeNetPart, SyntheticShapeNetPart
    d = tmp_path / "shapenet_part_seg_hdf5_data"
    d.mkdir()
    rng = np.random.default_rng(0)
    for name, n in (("ply_data_train0", 5), ("ply_data_val0", 3), ("ply_data_test0", 4)):
        label = rng.integers(0, 16, (n, 1))
        pid = np.asarray(INDEX_START)[label] + rng.integers(0, 2, (n, 64))
        np.savez(d / f"{name}.npz", data=rng.standard_normal((n, 64, 3)).astype(np.float32), label=label.astype(np.uint8), pid=pid.astype(np.uint8))
    tv = ShapeNetPart(32, "trainval", None, batch_size=4, shuffle=True, root=str(tmp_path))
    assert tv.data.shape == (8, 64, 3) and len(tv) == 2 and tv.seg_num_all == 50 and tv.seg_start_index == 0
    batches = list(tv)
    assert len(batches) == 2
    pts, label, seg = batches[0]
    assert pts.shape == (4, 32, 3) and pts.dtype.is_floating_point and label.shape == (4, 1) and seg.shape == (4, 32)
    assert label.dtype == seg.dtype and str(seg.dtype) == "torch.int64"
    te = ShapeNetPart(32, "test", None, batch_size=4, root=str(tmp_path))
    p0, l0, s0 = te[0]
    np.testing.assert_array_equal(p0, te.data[0][:32])          # no shuffle outside 'trainval'
    q0, _, t0 = tv[0]
    assert sorted(map(tuple, q0.tolist())) == sorted(map(tuple, tv.data[0][:32].tolist()))     # a permutation of the first 32
    cid = int(te.label[0, 0])
    cat = [k for k, v in te.cat2id.items() if v == cid][0]
    one = ShapeNetPart(32, "test", cat, batch_size=1, root=str(tmp_path))
    assert one.seg_num_all == SEG_NUM[cid] and one.seg_start_index == INDEX_START[cid] and (one.label == cid).all()
    import pytest
    with pytest.raises(FileNotFoundError):
        ShapeNetPart(32, "test", root=str(tmp_path / "nowhere"))
    syn = SyntheticShapeNetPart(128, "trainval", None, batch_size=4, n_items=8)
    pts, label, seg = next(iter(syn))
    lo = np.asarray(INDEX_START)[label.numpy()]
    assert pts.shape == (4, 128, 3) and ((seg.numpy() >= lo) & (seg.numpy() < lo + np.asarray(SEG_NUM)[label.numpy()])).all()


def test_balanced_accuracy_matches_sklearn():
    import importlib.util, os
    import numpy as np
    from sklearn import metrics
    spec = importlib.util.spec_from_file_location("train_partseg", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "train_partseg.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rng = np.random.default_rng(1)
    t, p = rng.integers(0, 50, 5000), rng.integers(0, 50, 5000)
    p[:2000] = t[:2000]
    assert abs(m.balanced_accuracy(t, p, 50) - metrics.balanced_accuracy_score(t, p)) < 1e-12
