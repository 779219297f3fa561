"""Training-step pieces of the reference's classification loop (train_cls.py) on PyTorch.

* ``soft_cross_entropy_loss`` -- train_cls.py:31-51 (label smoothing eps = 0.2); the reference builds the
  one-hot with a per-sample host sync (:41-42), here it is a device-side scatter.
* ``make_sgd`` -- ``nn.SGD(net.parameters(), lr, momentum)`` (train_cls.py:404): g += wd*p; v = mu*v + g;
  p -= lr*v, no dampening, no Nesterov == torch.optim.SGD.
* ``calculate_shape_IoU`` -- train_partseg.py:24-63 (per-shape mean part IoU over the parts of the shape's category; an
  empty union counts as IoU 1).
"""
import numpy as np
import torch
import torch.nn.functional as F

# ShapeNet-part: number of parts per object category and their first part id (train_partseg.py:24-25)
seg_num = [4, 2, 2, 4, 4, 3, 3, 2, 4, 2, 6, 2, 3, 3, 3, 3]
index_start = [0, 4, 6, 8, 12, 16, 19, 22, 24, 28, 30, 36, 38, 41, 44, 47]


_ONES = {}


def _one(device):
    """The scalar 1.0 on ``device`` that ``loss_backward`` seeds autograd with (one tensor per device, created once)."""
    t = _ONES.get(device)
    if t is None:
        t = _ONES[device] = torch.ones((), dtype=torch.float32, device=device)
    return t


def loss_backward(loss):
    """``loss.backward()`` for a scalar loss without its two smallest launches: autograd is seeded with a cached device scalar 1.0
    instead of a freshly filled ``ones_like(loss)``, and the loss kernels' backward, which already hold d loss / d logits, hand it on
    as it is when the incoming gradient IS that cached 1.0 (recognised by its address: no read of a device value, no synchronisation)
    instead of multiplying by it.  Same gradients bit for bit (x * 1.0 == x); any other gradient takes the product as before.
    (The reference's ``optimizer.step(loss)``, train_cls.py:404, is backward + update of one scalar loss as well.)"""
    torch.autograd.backward(loss, grad_tensors=(_one(loss.device),))


def _scaled(dx, g):
    one = _ONES.get(g.device)
    if one is not None and g.data_ptr() == one.data_ptr():
        return dx
    return dx * g


class _SoftCE(torch.autograd.Function):
    """pcl_soft_ce_f32: the loss and its gradient from one launch."""

    @staticmethod
    def forward(ctx, output, target, eps):
        from . import _lib
        R, C = output.shape
        x = output.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        _lib.call("pcl_soft_ce_f32", x.data_ptr(), target.data_ptr(), float(eps), R, C, loss.data_ptr(),
                  None if dx is None else dx.data_ptr(), torch._C._cuda_getCurrentRawStream(x.device.index))
        ctx.save_for_backward(dx)
        return loss

    @staticmethod
    def backward(ctx, g):
        dx, = ctx.saved_tensors
        return _scaled(dx, g), None, None


class _CERows(torch.autograd.Function):
    """pcl_soft_ce_rows_f32: mean cross entropy over the rows of a contiguous [R, C] matrix and its gradient, two launches."""

    @staticmethod
    def forward(ctx, rows, target, eps):
        from . import _lib
        R, C = rows.shape
        loss = torch.empty((), dtype=torch.float32, device=rows.device)
        partial = torch.empty(_lib.size_query("pcl_soft_ce_rows_blocks", R), dtype=torch.float32, device=rows.device)
        dx = torch.empty_like(rows) if ctx.needs_input_grad[0] else None
        _lib.call("pcl_soft_ce_rows_f32", rows.data_ptr(), target.data_ptr(), float(eps), R, C, partial.data_ptr(), loss.data_ptr(),
                  None if dx is None else dx.data_ptr(), torch._C._cuda_getCurrentRawStream(rows.device.index))
        ctx.save_for_backward(dx)
        return loss

    @staticmethod
    def backward(ctx, g):
        dx, = ctx.saved_tensors
        return _scaled(dx, g), None, None


def seg_cross_entropy_loss(scores, seg):
    """The part-segmentation loss of train_partseg.py:116, ``nn.cross_entropy_loss(pred, seg)``: mean cross entropy over every point.
    ``scores`` [B, part_num, N] as the networks return it (a transposed VIEW of the head's [B, N, part_num] rows) or [R, part_num];
    ``seg`` [B, N] / [R] integer part ids.  On the GPU the loss and its gradient are one library kernel + a fixed-order fold over the
    rows as the head wrote them -- PyTorch's composite is log-softmax forward / backward, three nll kernels and two layout copies of the
    [B, 50, N] tensor (81 us per PointNet++ part-seg step).  CPU tensors take ``F.cross_entropy`` (what the kernel is tested against)."""
    if scores.dim() == 3:
        rows = scores.permute(0, 2, 1)                     # [B, N, C]: the head's own layout when the network produced `scores`
        target = seg.reshape(-1)
    else:
        rows, target = scores, seg.reshape(-1)
    if not (scores.is_cuda and scores.dtype == torch.float32 and rows.numel() < 2 ** 31):
        return F.cross_entropy(scores, seg.long() if scores.dim() == 3 else target.long())
    rows = rows.reshape(-1, rows.shape[-1])                # a view when `rows` is contiguous; one copy otherwise
    if not rows.is_contiguous():
        rows = rows.contiguous()
    target = target.long()
    if target.device != rows.device:
        target = target.to(rows.device)
    return _CERows.apply(rows, target.contiguous(), 0.0)


def soft_cross_entropy_loss(output, target, smoothing=True):
    """train_cls.py:31-51 (eps = 0.2).  On the GPU the smoothed loss is one HIP kernel (forward + gradient); CPU tensors
    take the composite below, which is also what the kernel is tested against."""
    target = target.reshape(-1).long()
    if not smoothing:
        return F.cross_entropy(output, target)
    eps = 0.2
    if output.is_cuda and output.dtype == torch.float32 and output.shape[0] <= 65536:
        if target.device != output.device:          # a loader's CPU labels: the kernel would dereference a host pointer
            target = target.to(output.device)
        return _SoftCE.apply(output, target.contiguous(), eps)
    n_class = output.shape[1]
    one_hot = torch.zeros_like(output).scatter_(1, target[:, None], 1.0)
    one_hot = one_hot * (1 - eps) + (1 - one_hot) * eps / (n_class - 1)
    log_prb = F.log_softmax(output, dim=1)
    return -(one_hot * log_prb).sum(dim=1).mean()


class _LeanFusedSGD(torch.optim.SGD):
    """``torch.optim.SGD(fused=True)`` with the per-step Python bookkeeping (group walks, list building, hooks, profiler
    records: ~150 us a step for ~100 parameters, when the whole GPU step takes 2 ms) done once: after the first step the
    parameter and momentum-buffer lists are cached and ``step()`` is one call of the same multi-tensor kernel.  State,
    ``param_groups`` (learning-rate schedules) and ``state_dict`` stay those of the parent class."""

    def __init__(self, params, **kw):
        super().__init__(params, fused=True, **kw)
        self._lean = None
        self._own = None          # pointer tables of the library's own kernel (csrc/sgd.hip), built with the cached lists

    def load_state_dict(self, state_dict):
        # the parent replaces every momentum buffer with a new tensor: drop the cached lists, the next step() rebuilds them
        self._lean = self._own = None
        return super().load_state_dict(state_dict)

    def __setstate__(self, state):
        super().__setstate__(state)
        self._lean = self._own = None

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None or len(self.param_groups) != 1:
            return super().step(closure)
        g = self.param_groups[0]
        if self._lean is None:
            out = super().step()                                       # creates the momentum buffers
            ps = [p for p in g["params"] if p.grad is not None]
            if g["momentum"] != 0 and len(ps) == len(g["params"]) and len({(p.device, p.dtype) for p in ps}) == 1:
                bufs = [self.state[p]["momentum_buffer"] for p in ps]
                self._lean = (ps, bufs)
                self._own = None
                if ps and ps[0].dtype == torch.float32 and all(p.is_contiguous() for p in ps) and all(b.is_contiguous() for b in bufs):
                    import ctypes
                    n = len(ps)
                    U64, I64 = ctypes.c_uint64 * n, ctypes.c_int64 * n
                    self._own = (U64(*[p.data_ptr() for p in ps]), U64(*[b.data_ptr() for b in bufs]), I64(*[p.numel() for p in ps]), U64(), n)
            return out
        ps, bufs = self._lean
        # optimizer.state cleared or re-created behind our back: the cached buffers would be stale (first + last identity check)
        st0, st1 = self.state.get(ps[0]), self.state.get(ps[-1])
        if st0 is None or st1 is None or st0.get("momentum_buffer") is not bufs[0] or st1.get("momentum_buffer") is not bufs[-1]:
            self._lean = self._own = None
            return self.step()
        grads = [p.grad for p in ps]
        if any(x is None for x in grads):
            return super().step()
        if self._own is not None and not g["nesterov"] and not g["maximize"]:
            # one launch of the library's own kernel per 96 tensors (csrc/sgd.hip: torch's arithmetic, bit-identical): the pointer tables of
            # the parameters and the momentum buffers are built once, the gradients' every step (they are new slices of the stacks' flat
            # gradient buffers each backward)
            from . import _lib
            pa, ba, na, ga, n = self._own
            ok = True
            for i, x in enumerate(grads):
                if x.dtype != torch.float32 or x.device != ps[i].device:
                    ok = False              # (an fp64 / foreign-device gradient: torch's kernel below takes or refuses it)
                    break
                if not x.is_contiguous():   # (same values, dense: torch's multi-tensor kernel refuses strided gradients altogether)
                    x = ps[i].grad = grads[i] = x.contiguous()
                ga[i] = x.data_ptr()
                pa[i] = ps[i].data_ptr()    # (re-read every step: `p.data = ...` / `.to()` re-seat a parameter's storage without telling anybody)
            if ok:
                _lib.call("pcl_sgd_momentum_f32", pa, ga, ba, na, n, float(g["lr"]), float(g["momentum"]), float(g["weight_decay"]),
                          float(g["dampening"]), torch._C._cuda_getCurrentRawStream(ps[0].device.index))
                return None
        torch._fused_sgd_(ps, grads, bufs, weight_decay=g["weight_decay"], momentum=g["momentum"], lr=g["lr"],
                          dampening=g["dampening"], nesterov=g["nesterov"], maximize=g["maximize"], is_first_step=False,
                          grad_scale=None, found_inf=None)
        return None


def make_sgd(params, lr=0.02, momentum=0.9, weight_decay=0.0):
    params = list(params)
    # one multi-tensor kernel for the whole update on the GPU (same arithmetic as the default three-kernel foreach path)
    if params and all(p.is_cuda for p in params):
        return _LeanFusedSGD(params, lr=lr, momentum=momentum, weight_decay=weight_decay)
    return torch.optim.SGD(params, lr=lr, momentum=momentum, weight_decay=weight_decay)


def calculate_shape_IoU(pred_np, seg_np, label, class_choice=None):
    """pred_np, seg_np [S,N] part ids, label [S,1] category ids -> list of per-shape IoUs (train_partseg.py:27-48)."""
    pred_np, seg_np, label = np.asarray(pred_np), np.asarray(seg_np), np.asarray(label)
    shape_ious = []
    for shape_idx in range(seg_np.shape[0]):
        if not class_choice:
            idx = int(label[shape_idx][0])
            parts = range(index_start[idx], index_start[idx] + seg_num[idx])
        else:
            parts = range(seg_num[int(label[0])])
        part_ious = []
        for part in parts:
            inter = np.sum(np.logical_and(pred_np[shape_idx] == part, seg_np[shape_idx] == part))
            union = np.sum(np.logical_or(pred_np[shape_idx] == part, seg_np[shape_idx] == part))
            part_ious.append(1 if union == 0 else inter / float(union))
        shape_ious.append(np.mean(part_ious))
    return shape_ious
