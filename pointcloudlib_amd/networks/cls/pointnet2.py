"""PointNet++ classification (SSG / MSG) -- counterpart of /root/reference/networks/cls/pointnet2.py.

Same module structure and constructor arguments as the reference (PointNetModuleBase :11-62,
PointnetModule :65-80, PointnetModuleMSG :83-97, PointNet2_cls :100-158, PointNetMSG :161-196), on
channel-last tensors end to end: the reference's two NCHW transposes around ``nn.Conv(k=1)``
(:53, :56) disappear because a 1x1 conv is a row-wise linear map (misc/layers.py: PointwiseMLP).

Deliberate deviations from upstream bugs: ``PointnetModuleMSG`` iterates the
``mlps`` list directly (upstream :96 calls ``.layers.items()`` on a Python list and cannot be built).
"""
import os
from typing import List, Optional

import torch
from torch import nn

from ...misc.head import fc_head
from ...misc.layers import PointwiseMLP
from ...misc.ops import (BALL_QUERY_MULTI_MAX, BallQueryGrouper, FurthestPointSampler, GroupAll, ball_query, ball_query_multi, group_offsets,
                         group_offsets_multi, group_points)


_MULTI_BALL_QUERY = os.environ.get("PCL_MULTI_BALL_QUERY", "1") != "0"        # lab switch for A/B timing on one box


class PointNetModuleBase(nn.Module):
    def __init__(self):
        super().__init__()
        self.n_points = None
        self.sampler = None
        self.groupers = None
        self.mlps = None
        self.compact_duplicates = True      # HIP path: skip the padded duplicates of ball-query groups (same results)

    def build_mlps(self, mlp_spec: List[int], use_xyz: bool = True, bn: bool = True) -> PointwiseMLP:
        spec = list(mlp_spec)           # upstream mutates the caller's list in place (:22-23); we copy
        if use_xyz:
            spec[0] += 3
        return PointwiseMLP(spec, bias=not bn, bn=bn, slope=0.0)

    def sample(self, xyz: torch.Tensor):
        """The index-producing half of ``forward`` (no gradients, depends on xyz only): FPS centres and the
        ball-query neighbour lists of every grouper.  Returns (new_xyz, [idx per grouper])."""
        if self.n_points is None:
            return None, [None] * len(self.groupers)
        new_xyz = self.sampler(xyz)                                             # :45
        out = []
        gs = list(self.groupers)
        if 1 < len(gs) <= BALL_QUERY_MULTI_MAX and _MULTI_BALL_QUERY:
            # multi-scale grouping: every scale's list from one scan of the cloud (same lists as one ball_query per scale)
            lists = ball_query_multi(new_xyz, xyz, [g.radius for g in gs], [g.n_samples for g in gs], return_cnt=True)
        else:
            lists = [ball_query(new_xyz, xyz, g.radius, g.n_samples, return_cnt=True) for g in gs]
        if self.compact_duplicates and 1 < len(gs) <= BALL_QUERY_MULTI_MAX and _MULTI_BALL_QUERY:
            offs = group_offsets_multi([cnt for _, cnt in lists])
        else:
            offs = [group_offsets(cnt) if self.compact_duplicates else None for _, cnt in lists]
        for (idx, cnt), off in zip(lists, offs):
            out.append((idx, cnt, off))
        return new_xyz, out

    def forward(self, xyz: torch.Tensor, feature: Optional[torch.Tensor], sampling=None):
        """xyz [B,N,3], feature [B,N,C] -> (new_xyz [B,n_points,3] | None, new_feature [B,n_points,C']).
        ``sampling`` = a precomputed result of ``sample(xyz)`` (see PointNet2_cls.precompute_sampling)."""
        new_xyz, idxs = sampling if sampling is not None else self.sample(xyz)
        new_feature_list = []
        for grouper, mlp, ic in zip(self.groupers, self.mlps, idxs):
            if ic is None:
                grouped = grouper(new_xyz, xyz, feature)                        # GroupAll
                new_feature_list.append(mlp(grouped, group_max=grouped.shape[2]))
            elif self.compact_duplicates and mlp.resolved_backend(xyz) == "hip":
                # ball-query padding repeats the first hit: run the MLP on the distinct rows only (see ops.RowSet)
                new_feature_list.append(mlp.forward_grouped(xyz, new_xyz, feature, ic[0], ic[1], ic[2], grouper.use_xyz))
            else:
                grouped = group_points(xyz, new_xyz, feature, ic[0], grouper.use_xyz)     # [B, m, ns, C]   :51
                new_feature_list.append(mlp(grouped, group_max=grouped.shape[2]))       # conv/bn/relu x3 + max :54-57
        new_feature = new_feature_list[0] if len(new_feature_list) == 1 else torch.cat(new_feature_list, dim=-1)
        return new_xyz, new_feature

    def execute(self, *a, **k):
        return self(*a, **k)


class PointnetModule(PointNetModuleBase):
    def __init__(self, mlp: List[int], n_points=None, radius=None, n_samples=None, bn=True, use_xyz=True):
        super().__init__()
        self.n_points = n_points
        self.groupers = nn.ModuleList()
        if self.n_points is not None:
            self.sampler = FurthestPointSampler(n_points)
            self.groupers.append(BallQueryGrouper(radius, n_samples, use_xyz))
        else:
            self.groupers.append(GroupAll(use_xyz))
        self.mlps = nn.ModuleList()
        self.mlps.append(self.build_mlps(mlp, use_xyz, bn))


class PointnetModuleMSG(PointNetModuleBase):
    def __init__(self, n_points: int, radius: List[float], n_samples: List[int], mlps: List[List[int]], bn=True,
                 use_xyz=True):
        super().__init__()
        self.n_points = n_points
        self.sampler = FurthestPointSampler(n_points)
        self.groupers = nn.ModuleList()
        for r, s in zip(radius, n_samples):
            self.groupers.append(BallQueryGrouper(r, s, use_xyz))
        self.mlps = nn.ModuleList()
        for mlp in mlps:
            self.mlps.append(self.build_mlps(mlp, use_xyz, bn))


_AB_RECORD_STREAM = bool(os.environ.get("PCL_AB_RECORD_STREAM"))        # lab switch for A/B timing on one box


_OWN_SAMPLING_STREAM = {}          # device index -> the library's private producer stream
_OWN_LAST_CONSUMER = {}            # device index -> the consumer stream the private stream was last fed from


def sampling_stream(owner, stream, device=None):
    """-> (stream, owned).  ``stream="own"`` selects the library's PRIVATE producer stream of the current device: created here on
    first use (high priority: the work is a short latency-bound chain), never handed to anybody else, so the only allocations and
    kernels it ever sees are those of ``precompute_sampling`` calls -- each of which starts by waiting for its consumer stream.  That
    ownership is what makes a handle's memory safe without ``Tensor.record_stream`` (``adopt_sampling``); a stream the caller supplies
    may carry other work, and handles produced on it take the ``record_stream`` route.  ONE stream per device, shared by
    all networks: HIP multiplexes streams onto a handful of hardware queues, and a process that created a fresh side stream per
    network found its fourth one on the main stream's queue -- the step went from 4.9 to 10 ms (round 4, bench.py's other_configs)."""
    if isinstance(stream, str):
        if stream != "own":
            raise ValueError(f"stream={stream!r}: pass a torch.cuda.Stream, None or 'own'")
        dev = torch.cuda.current_device() if device is None or device.index is None else device.index      # (the tensor's device)
        s = _OWN_SAMPLING_STREAM.get(dev)
        if s is None:
            s = _OWN_SAMPLING_STREAM[dev] = torch.cuda.Stream(device=dev, priority=-1)
        return s, True
    return stream, False


class SamplingPrefetch:
    """For networks with a ``pointnet_modules`` list: every index-producing op (FPS + ball query per level) depends on xyz
    only, so the set for a batch can be produced ahead of its forward pass, on another stream."""

    def precompute_sampling(self, xyz, stream=None):
        """Run every index-producing op of the network (FPS + ball query per level: they depend on xyz only) for a
        batch, optionally on a side stream so that it overlaps other work -- the sampling of batch t+1 hides under
        the backward pass of batch t, the way an input pipeline would prepare it.  Returns a handle for
        ``forward(xyz, feature, sampling=handle)``.  The chain of m-1 dependent FPS steps occupies only B
        workgroups, so it costs nothing to run it beside the MFMA kernels."""
        cur = torch.cuda.current_stream()
        stream, owned = sampling_stream(self, stream, xyz.device)
        if stream is None:
            stream = cur
        if stream != cur:
            stream.wait_stream(cur)                       # xyz may have just been produced on the current stream
            if owned:
                # the private stream is shared by every network of the process: a handle produced for ANOTHER consumer stream may still be
                # read there, and this call is about to recycle its freed memory from the side pool -- wait for that consumer too
                # (the ordering argument of adopt_sampling needs every consumer the stream was fed from)
                dev = xyz.device.index if xyz.device.index is not None else torch.cuda.current_device()
                last = _OWN_LAST_CONSUMER.get(dev)
                if last is not None and last != cur:
                    stream.wait_stream(last)
                _OWN_LAST_CONSUMER[dev] = cur
            # ... and may be a temporary of the caller (x.transpose(1, 2).contiguous()): it dies when this call returns, and the
            # consumer stream's allocator pool would hand its memory out again while the producer stream still reads it
            xyz.record_stream(stream)
        out = []
        with torch.cuda.stream(stream), torch.no_grad():
            for module in self.pointnet_modules:
                s = module.sample(xyz)
                out.append(s)
                if s[0] is not None:
                    xyz = s[0]
            ev = torch.cuda.Event()
            ev.record(stream)
        # "owned" + "fed_from": every batch of work on the network's private stream starts by waiting for this consumer stream (the
        # wait_stream above), so memory of this handle that the host frees after enqueueing its consumers can only be handed out
        # again -- by the per-stream pools of the caching allocator, to a LATER call of this function -- behind those consumers.
        return {"levels": out, "event": ev, "stream": stream, "fed_from": cur, "owned": owned}

    @staticmethod
    def adopt_sampling(sampling):
        """Make the current stream wait for a handle produced on another stream.  When the handle was produced by
        ``precompute_sampling(..., stream="own")`` -- on the network's private stream -- from this same consumer stream, reuse of
        its memory is already ordered (see there) and nothing else is needed; for a handle of any other origin (a stream the
        caller supplied, another consumer stream) the allocator is told about the use (``record_stream``).
        That fallback is not free: every recorded tensor costs an event record on the consumer stream when it dies -- ten
        marker packets = 45 us of idle main stream at the head of each PointNet++ step (tools/dbg/section_times.py)."""
        if sampling is None:
            return
        cur = torch.cuda.current_stream()
        if sampling.get("event") is not None and sampling["stream"] != cur:
            cur.wait_event(sampling["event"])
            if sampling.get("owned") and sampling.get("fed_from") == cur and not _AB_RECORD_STREAM:
                return
            for new_xyz, idxs in sampling["levels"]:              # allocator safety across streams
                for t in [new_xyz] + [u for ic in idxs if ic is not None for u in ic]:
                    if t is not None:
                        t.record_stream(cur)


class PointNet2_cls(SamplingPrefetch, nn.Module):
    def __init__(self, n_classes=40, use_xyz=True):
        super().__init__()
        self.n_classes = n_classes
        self.use_xyz = use_xyz
        self.build_model()

    def build_model(self):
        self.pointnet_modules = nn.ModuleList()
        self.pointnet_modules.append(PointnetModule(n_points=512, radius=0.2, n_samples=64, mlp=[3, 64, 64, 128],
                                                    use_xyz=self.use_xyz))                       # :111-119
        self.pointnet_modules.append(PointnetModule(n_points=128, radius=0.4, n_samples=64, mlp=[128, 128, 128, 256],
                                                    use_xyz=self.use_xyz))                       # :121-129
        self.pointnet_modules.append(PointnetModule(mlp=[256, 256, 512, 1024], use_xyz=self.use_xyz))   # :131-136
        self.fc_layer = nn.Sequential(                                                           # :138-147
            nn.Linear(1024, 512, bias=False), nn.BatchNorm1d(512), nn.ReLU(),
            nn.Linear(512, 256, bias=False), nn.BatchNorm1d(256), nn.ReLU(),
            nn.Dropout(0.5), nn.Linear(256, self.n_classes),
        )

    def forward(self, xyz, feature, sampling=None):
        self.adopt_sampling(sampling)
        for i, module in enumerate(self.pointnet_modules):
            xyz, feature = module(xyz, feature, None if sampling is None else sampling["levels"][i])
        feature = feature.squeeze(dim=1)                                                         # :157
        return fc_head(self.fc_layer, feature)

    def execute(self, *a, **k):
        return self(*a, **k)


class PointNetMSG(PointNet2_cls):
    def build_model(self):
        super().build_model()
        self.pointnet_modules = nn.ModuleList()
        self.pointnet_modules.append(PointnetModuleMSG(
            n_points=512, radius=[0.1, 0.2, 0.4], n_samples=[16, 32, 128],
            mlps=[[3, 32, 32, 64], [3, 64, 64, 128], [3, 64, 96, 128]], use_xyz=self.use_xyz))  # :165-173
        input_channels = 64 + 128 + 128
        self.pointnet_modules.append(PointnetModuleMSG(
            n_points=128, radius=[0.2, 0.4, 0.8], n_samples=[32, 64, 128],
            mlps=[[input_channels, 64, 64, 128], [input_channels, 128, 128, 256], [input_channels, 128, 128, 256]],
            use_xyz=self.use_xyz))                                                               # :175-187
        self.pointnet_modules.append(PointnetModule(mlp=[128 + 256 + 256, 256, 512, 1024], use_xyz=self.use_xyz))
