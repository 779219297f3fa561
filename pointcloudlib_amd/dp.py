"""Data parallelism over the clouds of a batch: one process per GPU, RCCL over xGMI.

The reference has no distributed code; the hot path shards naturally because every
index op and every group is per-cloud.  The only exchange is the weight gradient: all parameter gradients
live in ONE flat fp32 buffer (``param.grad`` are views into it), so a step issues a single
``all_reduce(sum)`` of ~5.9 MB (PointNet++ SSG) and one scale -- sized for xGMI's per-link rate (7 x ~153 GB/s
point-to-point) instead of many small NCCL-style buckets.  BatchNorm statistics stay per-rank (weak scaling,
per-GPU batch = the reference's batch; a synchronised BatchNorm for strong-scaling parity is not built).
"""
import torch
import torch.distributed as dist


class FlatBucketDP:
    """Gradient exchange for replicated parameters.

    ``zero_grad()`` drops the gradients (autograd then *moves* fresh gradient tensors in instead of launching one
    accumulate kernel per parameter); ``all_reduce()`` packs them into the flat bucket with one multi-tensor copy,
    all-reduces the bucket once, scales it, and re-points ``param.grad`` at views of the bucket.  Without an initialised
    process group nothing is copied or communicated at all."""

    def __init__(self, module, process_group=None, broadcast=True):
        self.module = module
        self.group = process_group
        self.active = dist.is_available() and dist.is_initialized()        # a 1-rank group still runs the collective
        self.world = dist.get_world_size(process_group) if self.active else 1
        self.params = [p for p in module.parameters() if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views = []
        o = 0
        for p in self.params:
            n = p.numel()
            self.views.append(self.flat[o:o + n].view_as(p))
            o += n
        if broadcast and self.active:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=0, group=self.group)

    @property
    def nbytes(self):
        return self.flat.numel() * 4

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def all_reduce(self):
        """Average the gradients over ranks through ONE flat all-reduce (sum, then scale by 1/world)."""
        if not self.active:
            return
        src, dst = [], []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        if self.world > 1:
            self.flat.mul_(1.0 / self.world)
        for p, v in zip(self.params, self.views):
            p.grad = v


    def all_reduce_into_grads(self):
        """Same exchange for gradients that must stay where they are (``param.grad`` tensors written in place by a
        captured HIP graph): pack, one all-reduce, scale, unpack."""
        if not self.active:
            return
        grads = [p.grad for p in self.params if p.grad is not None]
        views = [v for p, v in zip(self.params, self.views) if p.grad is not None]
        torch._foreach_copy_(views, grads)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        if self.world > 1:
            self.flat.mul_(1.0 / self.world)
        torch._foreach_copy_(grads, views)


def shard_batch(tensors, rank, world):
    """Split the leading (cloud) axis of each tensor into `world` equal contiguous shards."""
    out = []
    for t in tensors:
        B = t.shape[0]
        assert B % world == 0, f"batch {B} not divisible by world size {world}"
        n = B // world
        out.append(t[rank * n:(rank + 1) * n])
    return out
