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
