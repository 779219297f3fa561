"""CPU restatements of the remaining part-segmentation networks (DGCNN, PointNet, PointConv) -- TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py; PARITY UNPINNED like every network restatement here: Jittor cannot be imported, these follow the text
of the reference files).  PointCNN's part-seg restatement lives in cpu_pointcnn.py (``pointcnn_partseg``).

Each class is state_dict-compatible with its pointcloudlib_amd counterpart and evaluates the reference's formulation on plain
PyTorch-CPU ops in fp32, fp64 or fp64-with-fp32-storage (cpu_common.ParamBag), with the index ops from pcl_oracle.c:

* ``DGCNNPartSegCPU``  -- /root/reference/networks/seg/dgcnn_partseg.py:11-128: ``get_graph_feature`` :11-33 WITH the edge tensor
  (kNN in the current feature space, k = 40), conv1-2 / conv3-4 / conv5 + max over k :95-107, concat :109, conv6 + max over the
  points :111-112, the 16-way label through conv7 :114-115, repeat + concat :117-120, conv8-11 :122-128.
* ``PointNetPartSegCPU`` -- networks/seg/pointnet_partseg.py:38-67 with the T-Nets of misc/layers.py:11-87.
* ``PointConvPartSegCPU`` -- networks/seg/pointconv_partseg.py:9-63: four ``PointConvDensitySetAbstraction`` levels
  (misc/pointconv_utils.py:340-400, restated in cpu_pointconv.py) and four ``PointConvDensitySetInterpolation`` levels
  (:252-330: 3-NN inverse-distance interpolation, then a PointConv over the N points themselves in FPS visiting order).
Dropout is skipped on both sides in parity tests.
"""
import numpy as np
import torch

from . import oracle as _o
from .cpu_common import ParamBag
from .cpu_pointconv import PointConvClsCPU


class DGCNNPartSegCPU(ParamBag):
    """state_dict-compatible with pointcloudlib_amd.networks.seg.dgcnn_partseg.DGCNN_partseg."""

    def __init__(self, state, k=40, dtype=torch.float32, storage=None):
        super().__init__(state, dtype, storage)
        self.k = k

    def knn_lists(self, x):
        xt = np.ascontiguousarray(x.detach().float().numpy().transpose(0, 2, 1))
        return torch.from_numpy(_o.knn(xt, xt, self.k).transpose(0, 2, 1).astype(np.int64))      # dgcnn_partseg.py:15-17

    def edge(self, x, idx):
        B, N, C = x.shape
        nb = x[torch.arange(B)[:, None, None], idx]                                # :26-27
        ctr = x[:, :, None, :].expand(B, N, self.k, C)                             # :28
        return torch.cat([self.rs(nb - ctr), ctr], dim=-1).reshape(-1, 2 * C)      # :30

    def forward(self, x, l, lists=None, return_aux=False):
        """x [B,3,N], l one-hot [B,16] -> [B,part_num,N] (aux: x1..x3 [B,N,64] and the neighbour lists used)."""
        x = x.to(self.dtype).transpose(1, 2)
        l = l.to(self.dtype)
        B, N, _ = x.shape
        feats, used, cur = [], [], x
        for s, name in enumerate(("conv12", "conv34", "conv5")):                   # :95-107
            idx = self.knn_lists(cur) if lists is None else lists[s]
            cur = self.mlp(f"{name}.", self.edge(cur, idx), slope=0.2).reshape(B, N, self.k, -1).max(dim=2)[0]
            feats.append(cur); used.append(idx)
        x123 = torch.cat(feats, dim=2)                                             # :109
        g = self.mlp("conv6.", x123.reshape(B * N, -1), slope=0.2).reshape(B, N, -1).max(dim=1)[0]    # :111-112
        lf = self.mlp("conv7.", l.reshape(B, 16), slope=0.2)                       # :114-115
        glob = torch.cat((g, lf), dim=1)[:, None, :].expand(B, N, 1088)            # :117-118
        y = torch.cat((glob, x123), dim=2).reshape(B * N, -1)                      # :120
        y = self.mlp("conv8.", y, slope=0.2)
        y = self.mlp("conv9.", y, slope=0.2)
        y = self.mlp("conv10.", y, slope=0.2)
        out = self.rs(torch.nn.functional.linear(y, self.g("conv11.weight"))).reshape(B, N, -1).permute(0, 2, 1)    # :127
        return (out, {"feats": feats, "lists": used}) if return_aux else out


class PointNetPartSegCPU(ParamBag):
    """state_dict-compatible with pointcloudlib_amd.networks.seg.pointnet_partseg.PointNet_partseg."""

    def stn(self, name, x):
        """misc/layers.py:28-56 / :75-87 on channel-last x [B,N,k] -> [B,k,k]"""
        B, N, k = x.shape
        y = self.mlp(f"{name}.convs.", x.reshape(B * N, k)).reshape(B, N, -1).max(dim=1)[0]
        y = self.mlp(f"{name}.fcs.", y)
        t = self.rs(torch.nn.functional.linear(y, self.g(f"{name}.fc3.weight"), self.g(f"{name}.fc3.bias")))
        return self.rs(t + torch.eye(k, dtype=self.dtype).reshape(1, k * k)).reshape(B, k, k)

    def forward(self, point_cloud, label, return_aux=False):
        """point_cloud [B,3,N], label one-hot [B,16] -> [B,part_num,N]"""
        pc = point_cloud.to(self.dtype).transpose(1, 2)
        label = label.to(self.dtype)
        B, N, _ = pc.shape
        pc = self.rs(torch.bmm(pc, self.stn("stn", pc)))                           # :41-43
        rows = lambda t: t.reshape(B * N, -1)
        out1 = self.mlp("conv1.", rows(pc))                                        # :47
        out2 = self.mlp("conv2.", out1)
        out3 = self.mlp("conv3.", out2)
        o3 = out3.reshape(B, N, -1)
        nt = self.rs(torch.bmm(o3, self.stn("fstn", o3)))                          # :51-54
        out4 = self.mlp("conv4.", rows(nt))                                        # :56
        out5 = self.mlp("conv5.", out4, last_act=False)                            # bn5(conv5(.)) :57
        out_max = out5.reshape(B, N, -1).max(dim=1)[0]                             # :58-59
        expand = torch.cat((out_max, label), 1)[:, None, :].expand(B, N, 2048 + 16)   # :61-62
        cat = torch.cat([expand] + [t.reshape(B, N, -1) for t in (out1, out2, out3, out4, out5)], 2)    # :63
        net = self.mlp("convs.", rows(cat))                                        # :64-66
        out = self.rs(torch.nn.functional.linear(net, self.g("convs4.weight"), self.g("convs4.bias")))  # :67
        out = out.reshape(B, N, -1).permute(0, 2, 1)
        return (out, {"pooled": out_max, "out3": o3}) if return_aux else out


class PointConvPartSegCPU(PointConvClsCPU):
    """state_dict-compatible with pointcloudlib_amd.networks.seg.pointconv_partseg.PointConvDensity_partseg."""

    SA = [("sa0", 1024, 32, 0.1), ("sa1", 256, 32, 0.2), ("sa2", 64, 32, 0.4), ("sa3", 36, 32, 0.8)]
    IN = [("in0", 16, 0.8), ("in1", 16, 0.4), ("in2", 16, 0.2), ("in3", 16, 0.1)]

    def interpolation(self, name, nsample, bandwidth, xyz1, xyz2, points2, start_idx, aux):
        """misc/pointconv_utils.py:275-323: xyz1 [B,N,3], xyz2 [B,S,3], points2 [B,S,D] -> [B,N,D'] (rows in FPS visiting order)"""
        B, N, _ = xyz1.shape
        x1 = np.ascontiguousarray(xyz1.detach().float().numpy())
        idx3, w3 = _o.three_nn(x1, np.ascontiguousarray(xyz2.detach().float().numpy()))              # :293-299
        nb = points2[torch.arange(B)[:, None, None], torch.from_numpy(idx3.astype(np.int64))]
        interp = self.rs((nb * torch.from_numpy(w3).to(self.dtype)[..., None]).sum(dim=2))            # :300
        aux.append({"three_nn": idx3})
        # sample_and_group(N, nsample, xyz1, interpolated, density) :307 == a set-abstraction level with npoint = N
        _, out = self.set_abstraction(name, N, nsample, bandwidth, xyz1, interp, start_idx, aux)
        return out

    def forward(self, xyz, start_idx, return_aux=False):
        """xyz [B,N,3]; start_idx: eight int arrays [B] -- the FPS starts of sa0..sa3, in0..in3 in call order -> [B,N,part_num]"""
        xyz0 = xyz.to(self.dtype)
        B, N, _ = xyz0.shape
        aux, lv = [], [(xyz0, None)]
        cur_xyz, points = xyz0, None
        for (name, npoint, nsample, bw), st in zip(self.SA, start_idx[:4]):                          # :44-47
            cur_xyz, points = self.set_abstraction(name, npoint, nsample, bw, cur_xyz, points, st, aux)
            lv.append((cur_xyz, points))
        p = lv[4][1]
        for j, ((name, ns, bw), st) in enumerate(zip(self.IN, start_idx[4:])):                       # :51-54
            fine = lv[3 - j][0]
            coarse = lv[4 - j][0]
            p = self.interpolation(name, ns, bw, fine, coarse, p, st, aux)
            # (the interpolation level emits its rows in FPS visiting order and the next level pairs them with ``fine``'s
            #  coordinates in the ORIGINAL order, as upstream does: kept as written)
        x = self.mlp("fc1.", p.reshape(B * N, -1))                                                   # :58
        out = self.rs(torch.nn.functional.linear(x, self.g("fc3.weight"), self.g("fc3.bias"))).reshape(B, N, -1)   # :59-61
        return (out, aux) if return_aux else out
