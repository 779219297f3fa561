"""Time pcl_fps_f32 (HIP events) for one shape; thread count / issue priority via PCL_FPS_THREADS / PCL_FPS_PRIO (-> pcl_set_fps_tuning)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudlib_amd import synth
from pointcloudlib_amd.misc import ops
from pointcloudlib_amd import _lib
_lib.lib().pcl_set_fps_tuning(int(os.environ.get("PCL_FPS_THREADS", "0")), int(os.environ.get("PCL_FPS_PRIO", "3")))
B, N, m = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.from_numpy(synth.gauss_ball(B, N, 1)).cuda()
for _ in range(3):
    ops.furthest_point_sample(x, m)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    ops.furthest_point_sample(x, m)
e.record(); torch.cuda.synchronize()
t = s.elapsed_time(e) / 20
print(f"T={os.environ.get('PCL_FPS_THREADS','auto'):>5} B={B} N={N} m={m}: {t*1e3:8.1f} us  {t*1e3/(m-1)*1e3:7.1f} ns/step")
