"""CPU restatement of the PointNet++ SSG classification step -- test infrastructure / CPU baseline only.

Index ops (FPS, ball query) come from ``pcl_oracle.c`` (OpenMP over clouds); gathers, the 1x1-conv +
BatchNorm + ReLU stacks, the max over the group, the FC head, the label-smoothed loss and SGD are plain
PyTorch-CPU fp32 -- the reference's composition at networks/cls/pointnet2.py:33-62, :149-158 and
train_cls.py:31-75.  Labelled everywhere as "CPU restatement of reference semantics (Jittor not runnable)".
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as _o


def _bn_train(y, gamma, beta, eps=1e-5):
    return F.batch_norm(y, None, None, gamma, beta, True, 0.0, eps)


def sa_module_cpu(xyz, feat, weights, gammas, betas, n_points, radius, n_samples, tie_stride, return_aux=False):
    """xyz [B,N,3] (torch cpu), feat [B,N,C] -> (new_xyz, new_feat[B,m,Cout]); weights: list of [Cout,Cin]."""
    B, N, _ = xyz.shape
    xyz_np = np.ascontiguousarray(xyz.detach().float().numpy())      # index ops are fp32 by definition (misc/ops.py CUDA text)
    if n_points is not None:
        fidx, new_xyz_np = _o.fps(xyz_np, n_points, block_size=tie_stride, return_xyz=True)
        idx = _o.ball_query(new_xyz_np, xyz_np, radius, n_samples)
        new_xyz = torch.from_numpy(new_xyz_np).to(xyz.dtype)
        bi = torch.arange(B)[:, None, None]
        li = torch.from_numpy(idx.astype(np.int64))
        g_xyz = xyz[bi, li] - new_xyz[:, :, None, :]
        grouped = torch.cat([g_xyz, feat[bi, li]], -1) if feat is not None else g_xyz
    else:
        fidx = idx = None
        new_xyz = None
        grouped = torch.cat([xyz, feat], -1)[:, None]
    lead = grouped.shape[:-1]
    y = grouped.reshape(-1, grouped.shape[-1])
    for w, g, b in zip(weights, gammas, betas):
        y = F.relu(_bn_train(F.linear(y, w), g, b))
    y = y.reshape(*lead, -1).max(dim=2)[0]
    if return_aux:
        return new_xyz, y, {"fps_idx": fidx, "bq_idx": idx, "grouped": grouped}
    return new_xyz, y


class PointNet2ClsCPU(torch.nn.Module):
    """Holds a state_dict-compatible copy of pointcloudlib_amd.networks.cls.pointnet2.PointNet2_cls."""

    SA = [(512, 0.2, 64), (128, 0.4, 64), (None, None, None)]

    def __init__(self, gpu_model_state, n_classes=40, tie_stride=8, dtype=torch.float32):
        """``dtype=torch.float64``: the same composition with every dense op in double precision (index ops stay fp32) --
        the ground truth the fp32 restatement AND the HIP path are both measured against in the gradient-parity tests."""
        super().__init__()
        self.tie_stride = tie_stride
        self.dtype = dtype
        self.p = torch.nn.ParameterDict()
        self.keys = {}
        for k, v in gpu_model_state.items():
            if "running" in k or "num_batches" in k:
                continue
            nk = k.replace(".", "__")
            self.p[nk] = torch.nn.Parameter(v.detach().cpu().to(dtype).clone())
            self.keys[k] = nk

    def g(self, k):
        return self.p[self.keys[k]]

    def forward(self, xyz, feat, return_aux=False):
        xyz = xyz.to(self.dtype)
        feat = None if feat is None else feat.to(self.dtype)
        aux = []
        for i, (m, r, ns) in enumerate(self.SA):
            pre = f"pointnet_modules.{i}.mlps.0."
            ws = [self.g(pre + f"weights.{j}") for j in range(3)]
            gs = [self.g(pre + f"gammas.{j}") for j in range(3)]
            bs = [self.g(pre + f"betas.{j}") for j in range(3)]
            res = sa_module_cpu(xyz, feat, ws, gs, bs, m, r, ns, self.tie_stride, return_aux)
            new_xyz, feat = res[0], res[1]
            if return_aux:
                a = res[2]
                a["feat"] = feat
                aux.append(a)
            xyz = new_xyz if new_xyz is not None else xyz
        x = feat.squeeze(1)
        x = F.relu(_bn_train(F.linear(x, self.g("fc_layer.0.weight")), self.g("fc_layer.1.weight"), self.g("fc_layer.1.bias")))
        x = F.relu(_bn_train(F.linear(x, self.g("fc_layer.3.weight")), self.g("fc_layer.4.weight"), self.g("fc_layer.4.bias")))
        # Dropout(0.5) is skipped on BOTH sides in parity tests (eval-mode dropout); the timed baseline keeps it.
        if self.training and getattr(self, "use_dropout", False):
            x = F.dropout(x, 0.5, True)
        x = F.linear(x, self.g("fc_layer.7.weight"), self.g("fc_layer.7.bias"))
        return (x, aux) if return_aux else x
