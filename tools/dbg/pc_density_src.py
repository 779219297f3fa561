"""Where does the error of PointConv's DensityNet weight gradient come from?  The PointConv parity test with the HIP net's
density input replaced by the CPU oracle's (same values the fp32 / fp64 restatements read)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import oracle as _o
_o.build()
from pointcloudlib_amd.misc import pointconv_utils as pu
import test_parity_pointconv_gpu as T
mode = sys.argv[1] if len(sys.argv) > 1 else "hip"
if mode == "cpu":
    def cd(xyz, bandwidth):
        return torch.from_numpy(_o.density(xyz.detach().cpu().numpy(), bandwidth)).to(xyz.device)
    pu.compute_density = cd
try:
    T.test_pointconv_cls_b32_n1024(_o, torch.device("cuda"))
except AssertionError as e:
    print("ASSERT", str(e)[:300])
