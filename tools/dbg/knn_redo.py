import sys, torch, time
sys.path.insert(0, ".")
from pointcloudlib_amd import _lib
from pointcloudlib_amd.misc.ops import _p, _stream
torch.manual_seed(0)
for C, scale in ((64, 1.0), (128, 1.0), (64, 0.1)):
    B, N, k = 32, 1024, 20
    x = (torch.randn(B, C, N, device="cuda") * scale).contiguous()
    idx = torch.empty((B, k, N), dtype=torch.int32, device="cuda")
    nbytes = _lib.lib().pcl_knn_workspace_bytes(B, C, N, N, k)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    _lib.call("pcl_knn_f32", _p(x), _p(x), B, C, N, N, k, _p(idx), _p(ws), nbytes, _stream())
    torch.cuda.synchronize()
    print(C, scale, "redo blocks:", int(ws[:B * 32].sum().item()), "of", B * 32)
