"""CPU restatement of DGCNN classification (BASELINE configs[2]) -- TEST INFRASTRUCTURE / CPU BASELINE ONLY.

Follows /root/reference/networks/cls/dgcnn.py as written: ``get_graph_feature`` :29-50 (kNN in the CURRENT feature space
through ``KNN(k)`` = the CUDA kernel of misc/ops.py:429-552, restated in pcl_oracle.c; gather; concat(nbr - ctr, ctr)),
the four conv+BN+LeakyReLU(0.2) stages with max over the k neighbours :100-111, concat + Conv1d 512->1024 :112-113,
global max || global mean :114-116 and the FC head :117-121.  Unlike the HIP path the edge tensor [B,N,k,2C] IS formed here
and the 1x1 conv runs over all B*N*k edges, exactly like the reference.  Dropout is skipped on both sides in parity tests.

``lists``: optional neighbour lists [4][B,N,k] to use instead of this model's own kNN.  The kNN of stages 2-4 runs on
features that two fp32 pipelines reproduce only to ~1e-6, so a near-tie can order two neighbours differently and the stage
outputs then differ by O(1) in a few rows; whole-network comparisons therefore share one set of lists (the fp32
restatement's), while index parity of the HIP kNN is checked on bit-identical inputs (tests/test_parity_dgcnn_gpu.py).
"""
import numpy as np
import torch

from . import oracle as _o
from .cpu_common import ParamBag, act


class DGCNNCPU(ParamBag):
    """state_dict-compatible with pointcloudlib_amd.networks.cls.dgcnn.DGCNN."""

    def __init__(self, state, k=20, dtype=torch.float32, storage=None):
        super().__init__(state, dtype, storage)
        self.k = k

    def knn_lists(self, x):
        """x [B,N,C] (any dtype; the kernel is fp32 by definition) -> int64 [B,N,k]"""
        xt = np.ascontiguousarray(x.detach().float().numpy().transpose(0, 2, 1))
        return torch.from_numpy(_o.knn(xt, xt, self.k).transpose(0, 2, 1).astype(np.int64))      # dgcnn.py:34-35

    def stage(self, name, x, idx):
        B, N, C = x.shape
        nb = x[torch.arange(B)[:, None, None], idx]                                # [B,N,k,C]      :44-46
        ctr = x[:, :, None, :].expand(B, N, self.k, C)                             # :47
        e = torch.cat([self.rs(nb - ctr), ctr], dim=-1)                            # :49
        y = self.mlp(f"{name}.", e.reshape(-1, 2 * C), slope=0.2)                  # conv + bn + LeakyReLU(0.2)  :72-83
        return y.reshape(B, N, self.k, -1).max(dim=2)[0]                           # :102

    def forward(self, x, lists=None, return_aux=False):
        """x [B,3,N] -> logits [B,n_classes] (aux: stage outputs x1..x4 [B,N,C] and the neighbour lists used)."""
        x = x.to(self.dtype).transpose(1, 2)
        feats, used = [], []
        cur = x
        for s in range(4):
            idx = self.knn_lists(cur) if lists is None else lists[s]
            cur = self.stage(f"conv{s + 1}", cur, idx)
            feats.append(cur); used.append(idx)
        x = torch.cat(feats, dim=2)                                                # :112
        B, N, _ = x.shape
        x = self.mlp("conv5.", x.reshape(B * N, -1), slope=0.2).reshape(B, N, -1)   # :113
        x = torch.cat([x.max(dim=1)[0], self.rs(x.mean(dim=1))], dim=1)            # :114-116
        x = self.fc_bn_act(x, "linear1", "bn6", 0.2)                               # :117
        x = self.fc_bn_act(x, "linear2", "bn7", 0.2)                               # :119
        x = self.fc_bn_act(x, "linear3")                                           # :121
        return (x, {"feats": feats, "lists": used}) if return_aux else x
