"""Synchronised BatchNorm statistics for the data-parallel step (a G-rank run with SyncBN must equal
the 1-rank run on the concatenated batch).

The fused MLP path keeps BatchNorm's batch sums as fp64 partial rows ``[rows][2][C]`` = (sum y, sum y^2) forward and
(sum du, sum du*y) backward (csrc/mlp.hip); synchronising is summing those rows locally and all-reducing ``2*C`` doubles
per layer before ``pcl_bn_finalize_f32`` / ``pcl_bn_bwd_consts_f32`` -- no extra pass over the activations.  Shards are
equal (``dp.shard_batch``), so the global row count is ``world x`` the local one and needs no exchange.

Backward: with L = mean over ranks of the per-rank losses, a rank's ``dgamma``/``dbeta`` stay sums over ITS rows (the
gradient all-reduce of dp.py averages them), while the constants of ``dy = a*du - k1 - k2*(y - mean)`` come from the
GLOBAL sums of du and du*y over the global row count (du being the per-rank-loss gradient on every rank, the 1/world of
L is applied by the gradient average).

Off by default (the reference has no distributed code; per-rank statistics = its per-GPU batch).  ``enable()`` is called
by ``FlatBucketDP(sync_bn=True)`` and ``disable()`` by its ``close()``.  The switch is PROCESS-GLOBAL while it is on: every
training-mode BatchNorm of the library (fused MLP stacks, EdgeConv, the FC head, the PointCNN / torch-backend
``batch_norm_train`` path) then issues collectives, so every rank must run the same training-mode forwards in the same order
-- a train-mode forward on one rank only deadlocks.  Not for use inside a HIP-graph capture (blocking all-reduces).
Equal shards are assumed (``dp.shard_batch`` asserts them; ``PCL_SYNCBN_CHECK=1`` verifies the row counts with one more
all-reduce + host sync per BatchNorm).
"""
import os
import torch
import torch.distributed as dist

_STATE = {"on": False, "group": None, "world": 1}


def enable(process_group=None):
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("syncbn.enable(): torch.distributed is not initialised")
    _STATE.update(on=True, group=process_group, world=dist.get_world_size(process_group))


def disable():
    _STATE.update(on=False, group=None, world=1)


def active():
    return _STATE["on"] and _STATE["world"] > 1


def world():
    return _STATE["world"] if _STATE["on"] else 1


def group():
    return _STATE["group"]


_CHECK = os.environ.get("PCL_SYNCBN_CHECK", "0") == "1"


def _check_equal_counts(count, device):
    c = torch.tensor([float(count), -float(count)], dtype=torch.float64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.MAX, group=_STATE["group"])
    if c[0].item() != -c[1].item():
        raise RuntimeError(f"syncbn: ranks hold different row counts (this rank {count}); shards must be equal")


def reduce_rows(stats, rows, count):
    """Partial rows of this rank -> (global sums as ONE row [1,2,C] fp64, 1, global row count)."""
    if _CHECK:
        _check_equal_counts(count, stats.device)
    g = stats[:rows].sum(dim=0, keepdim=True)               # fixed order -> deterministic; fp64
    dist.all_reduce(g, op=dist.ReduceOp.SUM, group=_STATE["group"])
    return g, 1, count * _STATE["world"]


class _SyncBN1d(torch.autograd.Function):
    """Training-mode BatchNorm over the rows of y [R,N] of ALL ranks (the FC head: one row per cloud), with the stack's
    running-statistics rule (biased variance, r += (batch - r) * momentum).  Sums are exchanged in fp64."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, eps, momentum):
        W = _STATE["world"]
        Rg = y.shape[0] * W
        yd = y.double()
        sums = torch.stack([yd.sum(0), (yd * yd).sum(0)])
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=_STATE["group"])
        mean = sums[0] / Rg
        var = (sums[1] / Rg - mean * mean).clamp_min(0.0)
        invstd = torch.rsqrt(var + eps)
        xhat = ((yd - mean) * invstd).float()
        with torch.no_grad():
            if running_mean is not None:
                running_mean += (mean.float() - running_mean) * momentum
                running_var += (var.float() - running_var) * momentum
        ctx.save_for_backward(xhat, gamma, invstd.float())
        ctx.Rg = Rg
        return xhat * gamma + beta

    @staticmethod
    def backward(ctx, g):
        xhat, gamma, invstd = ctx.saved_tensors
        gd = g.double()
        local = torch.stack([gd.sum(0), (gd * xhat.double()).sum(0)])
        glob = local.clone()
        dist.all_reduce(glob, op=dist.ReduceOp.SUM, group=_STATE["group"])
        dy = (gamma * invstd).double() * (gd - glob[0] / ctx.Rg - xhat.double() * (glob[1] / ctx.Rg))
        return dy.float(), local[1].float(), local[0].float(), None, None, None, None


def batch_norm_1d(y, bn):
    """``bn`` (an nn.BatchNorm1d in training mode) applied to y [R,N] with statistics over every rank's rows."""
    momentum = 0.1 if bn.momentum is None else bn.momentum
    return _SyncBN1d.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, momentum)


def batch_norm_rows(y2d, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5):
    """Training-mode BatchNorm over the rows of y2d [P,C] of ALL ranks (the plain-PyTorch BatchNorm paths of the library:
    ``layers.batch_norm_train``, PointCNN's BatchNorm after an activation, the head with more than 64 rows)."""
    if _CHECK:
        _check_equal_counts(y2d.shape[0], y2d.device)
    return _SyncBN1d.apply(y2d, gamma, beta, running_mean, running_var, eps, momentum)
