"""PointConv classification -- counterpart of /root/reference/networks/cls/pointconv.py:8-34 (BASELINE config 5).
sa3 uses ``group_all=True``, which upstream cannot run (missing ``sample_and_group_all``); see pointconv_utils."""
import torch
from torch import nn

from ...misc.head import fc_head
from ...misc.pointconv_utils import PointConvDensitySetAbstraction
from .pointnet2 import SamplingPrefetch, sampling_stream


class PointConvDensityClsSsg(nn.Module):
    def __init__(self, n_classes=40):
        super().__init__()
        self.sa1 = PointConvDensitySetAbstraction(npoint=512, nsample=32, in_channel=3, mlp=[64, 64, 128], bandwidth=0.1, group_all=False)
        self.sa2 = PointConvDensitySetAbstraction(npoint=128, nsample=64, in_channel=128 + 3, mlp=[128, 128, 256], bandwidth=0.2, group_all=False)
        self.sa3 = PointConvDensitySetAbstraction(npoint=1, nsample=None, in_channel=256 + 3, mlp=[256, 512, 1024], bandwidth=0.4, group_all=True)
        self.fc1 = nn.Linear(1024, 512)
        self.bn1 = nn.BatchNorm1d(512)
        self.drop1 = nn.Dropout(0.4)
        self.fc2 = nn.Linear(512, 256)
        self.bn2 = nn.BatchNorm1d(256)
        self.drop2 = nn.Dropout(0.4)
        self.fc3 = nn.Linear(256, n_classes)
        self.relu = nn.ReLU()

    def precompute_sampling(self, xyz, stream=None):
        """Everything of a batch that depends on its coordinates only -- per level the kernel density, the FPS centres (random start
        like :88 of the reference's utils) and their k-NN groups -- optionally on a side stream, one batch ahead (the protocol of
        networks/cls/pointnet2.SamplingPrefetch: the producer stream waits for the consumer stream first, which also orders the
        reuse of the handle's memory).  xyz [B,3,N]; returns a handle for ``forward(xyz, sampling=handle)``."""
        cur = torch.cuda.current_stream()
        stream, owned = sampling_stream(self, stream)          # "own": the network's private producer stream (pointnet2.sampling_stream)
        stream = cur if stream is None else stream
        if stream != cur:
            stream.wait_stream(cur)
            xyz.record_stream(stream)                          # a caller's temporary must outlive the producer stream's reads (pointnet2.py)
        out = []
        with torch.cuda.stream(stream), torch.no_grad():
            pts = xyz.permute(0, 2, 1).contiguous()
            for sa in (self.sa1, self.sa2, self.sa3):
                lv = sa.sample(pts)
                out.append(lv)
                if lv[0] is not None:
                    pts = lv[0]
            ev = torch.cuda.Event()
            ev.record(stream)
        return {"levels": out, "event": ev, "stream": stream, "fed_from": cur, "owned": owned}

    def forward(self, xyz, start_idx=None, knn_lists=None, sampling=None):
        """xyz [B,3,N] (the reference permutes from [B,N,3] at :26; callers here pass [B,3,N] directly).
        ``knn_lists``: optional neighbour groups (int32 [B,512,32], [B,128,64]) replacing ``knn_point``'s.
        ``sampling``: a handle of ``precompute_sampling(xyz)``."""
        B = xyz.shape[0]
        k1, k2 = (None, None) if knn_lists is None else knn_lists
        SamplingPrefetch.adopt_sampling(sampling)
        lv = [None, None, None] if sampling is None else sampling["levels"]
        l1_xyz, l1_points = self.sa1(xyz, None, None if start_idx is None else start_idx[0], k1, lv[0])
        l2_xyz, l2_points = self.sa2(l1_xyz, l1_points, None if start_idx is None else start_idx[1], k2, lv[1])
        _, l3_points = self.sa3(l2_xyz, l2_points, sampling=lv[2])
        x = l3_points.reshape(B, 1024)
        # one call per direction (misc/head.py: fc_head -> pcl_fc_head_*_f32)
        return fc_head([self.fc1, self.bn1, self.relu, self.drop1, self.fc2, self.bn2, self.relu, self.drop2, self.fc3], x)

    def execute(self, *a, **k):
        return self(*a, **k)
