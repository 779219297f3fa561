"""CPU restatement of PointConv classification (BASELINE configs[4]) -- TEST INFRASTRUCTURE / CPU BASELINE ONLY.

Follows /root/reference/misc/pointconv_utils.py and networks/cls/pointconv.py as written:
``PointConvDensitySetAbstraction.execute`` :361-400 -- Gaussian KDE ``compute_density`` :174-184 -> ``DensityNet`` :186-218
(every layer Conv1d+BatchNorm1d+ReLU: the sigmoid branch at :213 is unreachable) -> ``sample_and_group`` :133-170 (FPS from a
caller-supplied start index :74-116, ``knn_point`` :120-131, gather, centre, concat, gathered density) -> Conv+BN+ReLU
stack :384-389 -> ``WeightNet`` on the local coordinates :391-392 -> density multiply + per-point (C x ns)(ns x 16) matmul
:393-394 -> Linear + BatchNorm1d + ReLU :395-397; network head networks/cls/pointconv.py:27-34.

Index ops come from pcl_oracle.c: FPS without the origin skip and with block size 1 (the NumPy-style loop of :74-116 has no
block reduction: ties go to the lowest index, as ``jt.argmax`` returns the first maximum).  k-NN groups, two definitions:
``knn="direct"`` (default) -- (direct-form squared distance, index), the library's documented definition;
``knn="matmul"`` -- the reference's own arithmetic, ``square_distance`` in matmul form ``-2ab + a^2 + b^2`` (:34-53) + the
first k of a stable ascending argsort (:16-32, :129-130), restated in ``pclo_knn_point_matmul_f32`` (Jittor's matmul /
argsort rounding and tie order are outside /root/reference, so that restatement fixes: fma dot over c ascending, stable
sort).  tests/test_parity_pointconv_gpu.py bounds how many groups differ between the two and runs the whole network, forward
and backward, on BOTH sets of groups.  ``sample_and_group_all`` is missing upstream (:380); the original PointConv
semantics are used (one group of all points, coordinates relative to the centroid), as in pointcloudlib_amd.
"""
import numpy as np
import torch

from . import oracle as _o
from .cpu_common import ParamBag


class PointConvClsCPU(ParamBag):
    """state_dict-compatible with pointcloudlib_amd.networks.cls.pointconv.PointConvDensityClsSsg."""

    SA = [("sa1", 512, 32, 0.1), ("sa2", 128, 64, 0.2), ("sa3", None, None, 0.4)]
    knn = "direct"

    def set_abstraction(self, name, npoint, nsample, bandwidth, xyz, points, start_idx, aux):
        """xyz [B,N,3], points [B,N,D] | None -> (new_xyz [B,S,3], new_points [B,S,C'])"""
        B, N, _ = xyz.shape
        xyz_np = np.ascontiguousarray(xyz.detach().float().numpy())
        dens = torch.from_numpy(_o.density(xyz_np, bandwidth)).to(self.dtype)                  # :376
        dscale = self.mlp(f"{name}.densitynet.mlp.", dens.reshape(B * N, 1)).reshape(B, N, 1)   # :377
        rec = {}
        if npoint is None:                                                                      # sample_and_group_all
            new_xyz = xyz.mean(dim=1, keepdim=True)
            new_xyz = self.rs(new_xyz)
            g_xyz = self.rs(xyz[:, None] - new_xyz[:, :, None, :])                              # [B,1,N,3]
            new_points = torch.cat([g_xyz, points[:, None]], -1) if points is not None else g_xyz
            g_dens = dscale[:, None]                                                            # [B,1,N,1]
            S, ns = 1, N
        else:
            fidx = _o.fps(xyz_np, npoint, block_size=1, skip=False, start_idx=start_idx)        # :145
            bi = torch.arange(B)[:, None]
            new_xyz = xyz[bi, torch.from_numpy(fidx.astype(np.int64))]                          # :149
            q_np = np.ascontiguousarray(new_xyz.detach().float().numpy())
            if self.knn == "matmul":
                idx = _o.knn_point_matmul(nsample, xyz_np, q_np)                                # :152 as written (:34-53, :120-131)
            else:
                idx = _o.knn(np.ascontiguousarray(q_np.transpose(0, 2, 1)), np.ascontiguousarray(xyz_np.transpose(0, 2, 1)),
                             nsample).transpose(0, 2, 1)                                        # :152, direct-form definition
            li = torch.from_numpy(np.ascontiguousarray(idx).astype(np.int64))
            b3 = torch.arange(B)[:, None, None]
            g_xyz = self.rs(xyz[b3, li] - new_xyz[:, :, None, :])                               # :156-157
            new_points = torch.cat([g_xyz, points[b3, li]], -1) if points is not None else g_xyz     # :158-162
            g_dens = dscale[b3, li]                                                             # :169
            S, ns = npoint, nsample
            rec["fps_idx"], rec["knn_idx"] = fidx, np.ascontiguousarray(idx)
        f = self.mlp(f"{name}.mlp.", new_points.reshape(B * S * ns, -1)).reshape(B, S, ns, -1)  # :384-389
        w = self.mlp(f"{name}.weightnet.mlp.", g_xyz.reshape(B * S * ns, 3)).reshape(B, S, ns, -1)   # :391-392
        out = self.rs(torch.matmul(self.rs(f * g_dens).transpose(2, 3), w)).reshape(B, S, -1)   # :393-394
        out = self.mlp(f"{name}.linear.", out.reshape(B * S, -1)).reshape(B, S, -1)             # :395-397
        rec["new_xyz"], rec["feat"] = new_xyz, out
        aux.append(rec)
        return new_xyz, out

    def forward(self, xyz, start_idx, return_aux=False):
        """xyz [B,3,N]; start_idx: two int arrays [B] (the FPS start of sa1 / sa2, drawn by np.random.randint upstream :88)"""
        xyz = xyz.to(self.dtype).permute(0, 2, 1)
        B = xyz.shape[0]
        aux, points = [], None
        for (name, npoint, nsample, bw), st in zip(self.SA, list(start_idx) + [None]):
            xyz, points = self.set_abstraction(name, npoint, nsample, bw, xyz, points, st, aux)
        x = points.reshape(B, 1024)
        x = self.fc_bn_act(x, "fc1", "bn1", 0.0)          # networks/cls/pointconv.py:30 (dropout skipped on both sides)
        x = self.fc_bn_act(x, "fc2", "bn2", 0.0)
        x = self.fc_bn_act(x, "fc3")
        return (x, aux) if return_aux else x
