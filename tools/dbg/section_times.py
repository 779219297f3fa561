"""GPU time of the sections of one PointNet++ step WITHOUT a profiler attached: timing events recorded on the main stream
between forward / sampling hand-off / loss+backward / optimizer, 40 steps back to back with the host far ahead.  Compare
with the busy time of the same sections in profiles/*_step_timeline.csv: the difference is GPU idle that is not the host's."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from pointcloudlib_amd.affinity import pin_to_gpu_node; pin_to_gpu_node(0)
from pointcloudlib_amd import synth
from pointcloudlib_amd.networks.cls.pointnet2 import PointNet2_cls
from pointcloudlib_amd.train_utils import soft_cross_entropy_loss, make_sgd
torch.manual_seed(0)
net = PointNet2_cls().cuda().train()
opt = make_sgd(net.parameters(), lr=0.02, momentum=0.9)
x = torch.from_numpy(synth.gauss_ball(32, 1024, 1)).cuda(); f = torch.from_numpy(synth.unit_normals(32, 1024, 2)).cuda()
y = torch.from_numpy(synth.labels(32, 40, 3)).cuda()
params = list(net.parameters())
EXP = os.environ.get("EXP", "")
side = torch.cuda.Stream(priority=0 if EXP == "noprio" else -1)
if EXP in ("nowait", "norecord"):
    from pointcloudlib_amd.networks.cls import pointnet2 as _p2
    def _adopt(sampling):
        if sampling is None: return
        cur = torch.cuda.current_stream()
        if EXP == "norecord":
            cur.wait_event(sampling["event"])
    _p2.SamplingPrefetch.adopt_sampling = staticmethod(_adopt)
pending = {}
MODE = sys.argv[1] if len(sys.argv) > 1 else "prefetch"
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
def step(i, rec):
    for p in params: p.grad = None
    e0 = ev() if rec else None
    samp = pending.pop(i, None) if MODE == "prefetch" else None
    out = net(x, f, sampling=samp)
    e1 = ev() if rec else None
    if MODE == "prefetch":
        pending[i + 1] = net.precompute_sampling(x, stream=side)
    loss = soft_cross_entropy_loss(out, y)
    loss.backward()
    e2 = ev() if rec else None
    opt.step()
    e3 = ev() if rec else None
    return (e0, e1, e2, e3)
import gc
it = 0
for _ in range(150): step(it, False); it += 1
torch.cuda.synchronize(); gc.collect(); gc.disable()
for rec in (True,):
    evs = []
    t0 = ev()
    for _ in range(40): evs.append(step(it, rec)); it += 1
    t1 = ev()
    torch.cuda.synchronize()
    print(f"mode {MODE} {EXP}: step {t0.elapsed_time(t1) / 40 * 1e3:.1f} us (with 4 event records per step)")
    fw = sorted(a.elapsed_time(b) for a, b, _, _ in evs); bw = sorted(b.elapsed_time(c) for _, b, c, _ in evs); op = sorted(c.elapsed_time(d) for _, _, c, d in evs)
    nx = sorted(evs[i][3].elapsed_time(evs[i + 1][0]) for i in range(len(evs) - 1))
    m = lambda v: v[len(v) // 2] * 1e3
    print(f"  forward {m(fw):.1f} us | loss+backward {m(bw):.1f} us | optimizer {m(op):.1f} us | opt end -> next forward start {m(nx):.1f} us")
# without events, the plain step time
t0 = ev()
for _ in range(40): step(it, False); it += 1
t1 = ev(); torch.cuda.synchronize()
print(f"mode {MODE} {EXP}: step {t0.elapsed_time(t1) / 40 * 1e3:.1f} us (no events inside)")
