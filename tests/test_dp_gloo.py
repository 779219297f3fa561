"""CPU, world_size = 2 over gloo: the data-parallel layer (flat gradient bucket + one all-reduce) reproduces the
single-process gradient of the concatenated batch, parameters stay replicated, and batches shard by cloud."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_model():
    import oracle.torch_backend  # noqa: F401  (registers the plain-PyTorch composite: these workers run on CPU tensors)
    from pointcloudlib_amd.misc.layers import PointwiseMLP
    torch.manual_seed(7)
    m = PointwiseMLP([5, 16, 12, 8], bias=True, bn=False, backend="torch")     # no BatchNorm: grads are batch-additive
    return m


def _worker(rank, world, port, out_q, in_place=False, bucket_bytes=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pointcloudlib_amd.dp import FlatBucketDP, shard_batch
    model = _make_model()
    if rank == 1:                                   # rank 1 starts from different weights: broadcast must fix it
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    dp = FlatBucketDP(model, bucket_bytes=bucket_bytes, overlap=bool(bucket_bytes))
    if bucket_bytes:                         # several buckets, sent from the autograd hooks during backward
        assert len(dp.buckets) >= 3 and sum(dp.bucket_nbytes) == dp.nbytes
        sent = []
        orig = dp._send
        dp._send = lambda b: (sent.append((dp._armed, dp.buckets.index(b))), orig(b))
    torch.manual_seed(0)
    x = torch.randn(8, 12, 5)
    y = torch.randn(8, 12, 8)
    (xs, ys) = shard_batch([x, y], rank, world)
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
    for _ in range(2):
        dp.zero_grad()
        ((model(xs) - ys) ** 2).mean().backward()
        if in_place:            # the variant for gradients that must stay where a captured HIP graph writes them
            before = [p.grad.data_ptr() for p in model.parameters()]
            dp.all_reduce_into_grads()
            assert before == [p.grad.data_ptr() for p in model.parameters()]
        else:
            if bucket_bytes:
                n_early = sum(1 for armed, _ in sent if armed)
                assert n_early >= len(dp.buckets) - 1, sent          # all but (at most) the tail went out during backward
                assert [b for _, b in sent] == sorted(b for _, b in sent)     # in gradient-ready order
                sent.clear()
            dp.all_reduce()
            assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(dp.params, dp.views))
        opt.step()
    out_q.put((rank, [p.detach().numpy().copy() for p in model.parameters()], dp.flat.numpy().copy(), dp.nbytes))
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("in_place,bucket_bytes", [(False, None), (True, None), (False, 256)])
def test_flat_bucket_dp_matches_single_process(in_place, bucket_bytes):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, in_place, bucket_bytes)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the full batch
    model = _make_model()
    torch.manual_seed(0)
    x = torch.randn(8, 12, 5)
    y = torch.randn(8, 12, 8)
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
    for _ in range(2):
        opt.zero_grad()
        ((model(x) - y) ** 2).mean().backward()
        opt.step()
    ref = [p.detach() for p in model.parameters()]
    for r in range(world):
        assert res[r][3] == sum(p.numel() for p in ref) * 4
        for a, b in zip(res[r][1], ref):
            assert torch.allclose(torch.from_numpy(a), b, rtol=1e-5, atol=1e-6)
    assert (res[0][2] == res[1][2]).all()               # identical averaged gradient buckets on both ranks


def test_shard_batch_by_cloud():
    from pointcloudlib_amd.dp import shard_batch
    x = torch.arange(24).reshape(8, 3)
    a, = shard_batch([x], 1, 4)
    assert a.tolist() == x[2:4].tolist()


def _syncbn_worker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pointcloudlib_amd import syncbn
    from pointcloudlib_amd.dp import shard_batch
    syncbn.enable()
    assert syncbn.active() and syncbn.world() == world
    torch.manual_seed(3)
    y = torch.randn(12, 7) * 3 + 5
    g = torch.randn(12, 7)
    bn = torch.nn.BatchNorm1d(7).train()
    with torch.no_grad():
        bn.weight.uniform_(-1, 1.5); bn.bias.uniform_(-0.5, 0.5)
    (ys, gs) = shard_batch([y, g], rank, world)
    ys = ys.clone().requires_grad_(True)
    out = syncbn.batch_norm_1d(ys, bn)
    out.backward(gs)
    # partial-row reduction used by the fused MLP path: [rows][2][C] fp64 -> one global row, global count
    stats = torch.arange(3 * 2 * 4, dtype=torch.float64).reshape(3, 2, 4) * (rank + 1)
    red, rows, cnt = syncbn.reduce_rows(stats, 2, 10)
    out_q.put((rank, out.detach().numpy().copy(), ys.grad.numpy().copy(), bn.weight.grad.numpy().copy(), bn.bias.grad.numpy().copy(),
               bn.running_mean.numpy().copy(), bn.running_var.numpy().copy(), red.numpy().copy(), rows, cnt))
    dist.barrier()
    dist.destroy_process_group()


def test_syncbn_pieces_match_the_whole_batch():
    """syncbn.batch_norm_1d on 2 shards == BatchNorm over the 12 rows (forward, input gradient; the weight gradients sum over the
    ranks to the whole-batch ones; biased running variance); reduce_rows sums the first `rows` partial rows over the ranks."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_syncbn_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(3)
    y = (torch.randn(12, 7) * 3 + 5).requires_grad_(True)
    g = torch.randn(12, 7)
    bn = torch.nn.BatchNorm1d(7).train()
    with torch.no_grad():
        bn.weight.uniform_(-1, 1.5); bn.bias.uniform_(-0.5, 0.5)
    ref = torch.nn.functional.batch_norm(y, None, None, bn.weight, bn.bias, True, 0.0, bn.eps)
    ref.backward(g)
    out = torch.cat([torch.from_numpy(res[0][1]), torch.from_numpy(res[1][1])])
    dx = torch.cat([torch.from_numpy(res[0][2]), torch.from_numpy(res[1][2])])
    assert torch.allclose(out, ref.detach(), rtol=1e-5, atol=1e-5) and torch.allclose(dx, y.grad, rtol=1e-5, atol=1e-5)
    assert torch.allclose(torch.from_numpy(res[0][3] + res[1][3]), bn.weight.grad, rtol=1e-5, atol=1e-5)
    assert torch.allclose(torch.from_numpy(res[0][4] + res[1][4]), bn.bias.grad, rtol=1e-5, atol=1e-5)
    mean, var = y.detach().mean(0), y.detach().var(0, unbiased=False)
    for r in range(world):
        assert torch.allclose(torch.from_numpy(res[r][5]), 0.1 * mean, rtol=1e-5, atol=1e-6)
        assert torch.allclose(torch.from_numpy(res[r][6]), 0.9 + 0.1 * var, rtol=1e-5, atol=1e-6)
        want = (torch.arange(24, dtype=torch.float64).reshape(3, 2, 4)[:2].sum(0, keepdim=True) * 3).numpy()
        assert (res[r][7] == want).all() and res[r][8] == 1 and res[r][9] == 20
