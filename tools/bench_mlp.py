"""Micro-benchmark of the fused MLP stack on one SA shape (for rocprofv3 --pmc runs).
    python tools/bench_mlp.py [sa1|sa2|sa3] [iters] [fwd|fwdbwd]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudlib_amd.misc.layers import PointwiseMLP  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "sa1"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mode = sys.argv[3] if len(sys.argv) > 3 else "fwdbwd"
spec, shape, ns = {"sa1": ([6, 64, 64, 128], (32, 512, 64, 6), 64),
                   "sa2": ([131, 128, 128, 256], (32, 128, 64, 131), 64),
                   "sa3": ([259, 256, 512, 1024], (32, 1, 128, 259), 128)}[which]
torch.manual_seed(0)
m = PointwiseMLP(spec).cuda().train()
x = torch.randn(*shape, device="cuda", requires_grad=(which != "sa1"))
for it in range(iters):
    out = m(x, group_max=ns)
    if mode == "fwdbwd":
        out.backward(torch.randn_like(out))
torch.cuda.synchronize()
print("done", which, iters, mode)
