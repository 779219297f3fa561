import sys, time, torch
sys.path.insert(0, ".")
from pointcloudlib_amd.misc import ops
torch.manual_seed(0)
def t(C, k, B=32, N=1024, reps=20):
    x = torch.randn(B, C, N, device="cuda")
    for _ in range(3): ops.knn_indices(x, x, k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): ops.knn_indices(x, x, k)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
for C in (3, 64, 128):
    print(C, " ".join(f"k={k}: {t(C, k):7.1f} us" for k in (1, 20, 40, 65)), flush=True)
