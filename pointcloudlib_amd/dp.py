"""Data parallelism over the clouds of a batch: one process per GPU, RCCL over xGMI.

The reference has no distributed code; the hot path shards naturally because every
index op and every group is per-cloud.  The only exchange is the weight gradient.  All parameter gradients
live in ONE flat fp32 buffer (``param.grad`` are views into it), laid out in the order backward produces them
(reverse registration order).  Default: one ``all_reduce(sum)`` of the whole buffer (5.9 MB for PointNet++ SSG) after
backward and one scale -- sized for xGMI (7 point-to-point links x ~153 GB/s per GPU; ring collectives are per-link
bound and latency-dominated at this size, so fewer and larger beats NCCL-style small buckets).

Optional (``bucket_bytes``, ``overlap=True``): the buffer is cut into buckets and a bucket's all-reduce is issued from
an autograd hook the moment its last gradient exists, on RCCL's own stream, beside the rest of backward (94 % of the
PointNet++ gradients belong to the classifier head and the GroupAll level, which backward finishes first).  Measured
on one MI355X under torchrun (world = 1, so only the issue cost shows): single bucket after backward +0.05-0.1 ms/step
over no exchange, two buckets after backward the same, two buckets with the early one issued during backward +0.25 ms -- the
early issue lands in the part of backward where the GPU runs short kernels (head, GroupAll level) and waits for the
host, so ProcessGroupNCCL's ~0.1 ms of host work per collective is fully exposed there, more than the ~0.1 ms an
8-rank 5.9 MB all-reduce costs when left exposed.  Hence overlap is off by default.
BatchNorm statistics stay per-rank by default (weak scaling, per-GPU batch = the reference's batch); ``sync_bn=True``
all-reduces the fp64 batch sums of every training-mode BatchNorm the library runs (pointcloudlib_amd/syncbn.py): the fused
MLP stacks, EdgeConv's BatchNorm (misc/edgeconv.py), the FC head (any row count), the wide PointConv linear (routed to the
fused MLP path), and the plain-PyTorch ``batch_norm_train`` path (PointCNN's BatchNorm after an activation, the ``torch``
backend).  ``nn.BatchNorm*`` modules called directly by a user model (not through ``head_layer`` / ``fc_head``) are NOT
synchronised; ``FlatBucketDP(sync_bn=True)`` raises at its first gradient exchange if a training-mode BatchNorm module of the
wrapped model was not normalised by a synchronised path during that step (see ``_unsynced_batchnorms``).  The G-rank step then equals the 1-rank step on the concatenated batch
(tests/test_syncbn_gpu.py: PointNet++ and DGCNN).  The switch is process-global while the wrapper lives: ``close()``
turns it off.
"""
import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ("params", "views", "flat", "pending", "work")

    def __init__(self):
        self.params, self.views, self.flat, self.pending, self.work = [], [], None, 0, None


class FlatBucketDP:
    """Gradient exchange for replicated parameters.

    ``zero_grad()`` drops the gradients (autograd then *moves* fresh gradient tensors in instead of launching one
    accumulate kernel per parameter) and arms the buckets; ``all_reduce()`` packs every bucket not sent yet into its
    slice of the flat buffer with one multi-tensor copy (parameters without a gradient count as zero), all-reduces it,
    waits, scales by 1/world and re-points ``param.grad`` at views of the flat buffer.  With ``overlap=True`` a bucket is
    sent from an autograd hook during ``backward()`` as soon as its gradients are complete, strictly in bucket order
    (then: one ``backward()`` per ``zero_grad()``, no gradient accumulation across backward calls).
    Without an initialised process group nothing is copied or communicated at all."""

    def __init__(self, module, process_group=None, broadcast=True, bucket_bytes=None, overlap=False, sync_bn=False):
        self.module = module
        self.group = process_group
        self.active = dist.is_available() and dist.is_initialized()        # a 1-rank group still runs the collective
        self.sync_bn = bool(sync_bn) and self.active
        self._bn_checked = not self.sync_bn
        if self.sync_bn:
            from . import syncbn
            syncbn.enable(process_group)
        self.world = dist.get_world_size(process_group) if self.active else 1
        self.overlap = overlap
        self.params = [p for p in module.parameters() if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        view_of, self.buckets, self._bucket_of = {}, [], {}
        cur, lo, o = _Bucket(), 0, 0
        for p in reversed(self.params):                                    # ~ the order backward finishes them
            n = p.numel()
            v = self.flat[o:o + n].view_as(p)
            view_of[id(p)] = v
            cur.params.append(p)
            cur.views.append(v)
            self._bucket_of[id(p)] = cur
            o += n
            if bucket_bytes and (o - lo) * 4 >= bucket_bytes:
                cur.flat = self.flat[lo:o]
                self.buckets.append(cur)
                cur, lo = _Bucket(), o
        if cur.params:
            cur.flat = self.flat[lo:o]
            self.buckets.append(cur)
        self.views = [view_of[id(p)] for p in self.params]
        self._armed, self._next = False, 0
        if self.active:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._on_grad)
            if broadcast:
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t.data, src=0, group=self.group)

    def close(self):
        """Switch the process-global SyncBN state off again (a wrapper with ``sync_bn=True`` turned it on)."""
        if self.sync_bn:
            from . import syncbn
            syncbn.disable()
            self.sync_bn = False

    @property
    def nbytes(self):
        return self.flat.numel() * 4

    @property
    def bucket_nbytes(self):
        return [b.flat.numel() * 4 for b in self.buckets]

    def zero_grad(self):
        for p in self.params:
            p.grad = None
        for b in self.buckets:
            b.pending, b.work = len(b.params), None
        self._armed, self._next = True, 0

    def _send(self, b):
        src, dst = [], []
        for p, v in zip(b.params, b.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)
        b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _on_grad(self, p):
        if not (self._armed and self.overlap):
            return
        self._bucket_of[id(p)].pending -= 1
        # strictly in bucket order, so every rank issues the same sequence of collectives whatever order autograd
        # happens to finish the gradients inside a layer
        while self._next < len(self.buckets) and self.buckets[self._next].pending == 0:
            self._send(self.buckets[self._next])
            self._next += 1

    def all_reduce(self):
        """Average the gradients over ranks: finish the bucketed all-reduces (sum), then scale by 1/world."""
        if not self.active:
            return
        self._armed = False
        for b in self.buckets:
            if b.work is None:
                self._send(b)
        for b in self.buckets:
            b.work.wait()
            b.work = None
        if self.world > 1:
            self.flat.mul_(1.0 / self.world)
        for p, v in zip(self.params, self.views):
            p.grad = v
        # (after the exchange: a failed check must not leave collectives issued and un-waited on this rank only -- ADVICE r4)
        if not self._bn_checked:
            self._check_batchnorms()

    def _check_batchnorms(self):
        """After the first forward + backward of a ``sync_bn=True`` wrapper: every training-mode BatchNorm module of the model must
        have gone through a synchronised path (see ``_unsynced_batchnorms``)."""
        self._bn_checked = True
        bad = _unsynced_batchnorms(self.module)
        if bad:
            raise RuntimeError("FlatBucketDP(sync_bn=True): these BatchNorm modules were not normalised by a synchronised path in the first "
                               f"step (called directly by the model, they keep per-rank statistics): {bad}; route them through "
                               "misc.head.head_layer / fc_head or PointwiseMLP")

    def all_reduce_into_grads(self):
        """Same exchange for gradients that must stay where they are (``param.grad`` tensors written in place by a
        captured HIP graph): pack, one all-reduce of the whole flat buffer, scale, unpack.  Use with ``overlap=False``
        (hooks must not issue collectives inside a capture)."""
        if not self.active:
            return
        self._armed = False
        grads = [p.grad for p in self.params if p.grad is not None]
        views = [v for p, v in zip(self.params, self.views) if p.grad is not None]
        torch._foreach_copy_(views, grads)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        if self.world > 1:
            self.flat.mul_(1.0 / self.world)
        torch._foreach_copy_(grads, views)
        if not self._bn_checked:
            self._check_batchnorms()


def _unsynced_batchnorms(module):
    """Names of the ``nn.BatchNorm*`` modules in training mode that no synchronised path of the library has touched so far.  The
    counterpart networks keep ``nn.BatchNorm1d`` modules only as parameter containers of the FC head; ``misc.head.head_layer``
    (which ``fc_head`` and the stack path fall back to when synchronised statistics are on) marks every BatchNorm module it
    normalises with ``_pcl_sync_routed``.  A module without the mark after a forward pass was called directly by the model
    (``self.bn1(x)``) and kept per-rank statistics.  Decided by what ran, not by what a module is called."""
    import torch.nn as nn
    return [name for name, m in module.named_modules()
            if isinstance(m, nn.modules.batchnorm._BatchNorm) and m.training and not getattr(m, "_pcl_sync_routed", False)]


def shard_batch(tensors, rank, world):
    """Split the leading (cloud) axis of each tensor into `world` equal contiguous shards."""
    out = []
    for t in tensors:
        B = t.shape[0]
        assert B % world == 0, f"batch {B} not divisible by world size {world}"
        n = B // world
        out.append(t[rank * n:(rank + 1) * n])
    return out
