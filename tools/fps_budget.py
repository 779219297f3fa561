"""Cycle budget of the FPS chain: per dependent step, where do the ~900-1100 shader cycles go?
    make -C pointcloudlib_amd/csrc EXP=6 && PCL_HIP_SO=$PWD/pointcloudlib_amd/libpcl_hip_exp6.so python tools/fps_budget.py
Lab build: s_memtime stamps between the phases of a step (thread 0 of cloud 0; every stamp is an s_waitcnt lgkmcnt(0) + s_memtime, so the
stamped chain runs a few % longer than the product's -- the last column is the product build's event-timed step for comparison)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudlib_amd import synth, _lib
from pointcloudlib_amd.misc import ops
L = _lib.lib()
names = ["centre broadcast (LDS read)", "distances + lane-local key max", "wave reduction (6 DPP stages + 2 readlanes)",
         "cross-wave exchange (slot write, barrier, slot reads, maxima)", "rank decode + index store"]
for B, N, m, thr in ((32, 1024, 512, 0), (32, 1024, 512, 64), (32, 1024, 512, 128), (32, 512, 128, 0), (32, 4096, 1024, 0), (16, 2048, 512, 0)):
    L.pcl_set_fps_tuning(thr, 3)
    x = torch.from_numpy(synth.gauss_ball(B, N, 1)).cuda()
    for _ in range(3):
        ops.furthest_point_sample(x, m)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.furthest_point_sample(x, m)
    e.record(); torch.cuda.synchronize()
    ns = s.elapsed_time(e) / 10 * 1e6 / (m - 1)
    line = f"B={B} N={N} m={m} threads/cloud={thr or 'default'}: {ns:7.1f} ns per dependent step (events, this build)"
    if hasattr(L, "pcl_lab_fps_read"):
        buf = (ctypes.c_longlong * 8)()
        assert L.pcl_lab_fps_read(buf) == 0
        steps = max(1, buf[5])
        line += f"; T x PPT = {buf[7] // 1000} x {buf[7] % 1000}; chain {buf[6] / steps:7.1f} cycles per step = " + " | ".join(f"{names[i].split(' (')[0]} {buf[i] / steps:6.1f}" for i in range(5))
    print(line, flush=True)
L.pcl_set_fps_tuning(0, 3)
