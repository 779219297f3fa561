"""Lab measurement: the headline step -- PointNet++ SSG cls, B=32, N=1024, fwd + bwd + SGD -- replayed as ONE
HIP graph launch of its main-stream work against the eager step, same box, same process, interleaved windows, with and without a
busy-wait injected into the host at the head of every step (is the step host-independent?).

    python tools/graph_replay.py [--steps 20] [--windows 5] [--host-delay-us 600]

Protocol = bench.py's: the FPS / ball-query indices of batch t+1 are produced eagerly on the library's side stream during step t
(every step runs one full set), the main stream's zero_grad + forward + loss + backward + SGD are what the graph holds.  The graph
reads its inputs and the sampling handle from static buffers (one 5 MB device copy per step brings the handle over).
NOT the product path: inside a graph the FC head's dropout seed is frozen at capture (every replay drops the same units) -- fine
for a timing, wrong for training -- and `bench.py` could not event-time a kernel inside a replay.  The result of this measurement
is recorded in DESIGN.md; rounds 1-2 had measured the replay 2-4 % SLOWER than eager launches at 115 dispatches per step.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointcloudlib_amd.affinity import pin_to_gpu_node  # noqa: E402
AFF = pin_to_gpu_node(0)
import torch  # noqa: E402


def _flat(sampling):
    return [t for new_xyz, idxs in sampling["levels"] for t in [new_xyz] + [u for ic in idxs if ic is not None for u in ic]
            if t is not None]


class GraphedStep:
    def __init__(self, net, loss_fn, opt, dp, batch):
        self.net, self.loss_fn, self.opt, self.dp = net, loss_fn, opt, dp
        self.inputs = [t.clone() for t in batch]
        with torch.no_grad():
            s0 = net.precompute_sampling(batch[0])
        flat = _flat(s0)
        assert all(t.element_size() == 4 for t in flat)
        n = sum(t.numel() for t in flat)
        dev = flat[0].device
        self.cur = torch.empty(n, dtype=torch.int32, device=dev)
        self.nxt = torch.empty(n, dtype=torch.int32, device=dev)
        self.cur_views, self.nxt_views = self._views(self.cur, flat), self._views(self.nxt, flat)
        torch._foreach_copy_(self.nxt_views, flat)
        self.cur.copy_(self.nxt)
        it = iter(self.cur_views)
        levels = []
        for new_xyz, idxs in s0["levels"]:
            nx = next(it) if new_xyz is not None else None
            levels.append((nx, [None if ic is None else tuple(next(it) if u is not None else None for u in ic) for ic in idxs]))
        self.samp = {"levels": levels, "event": None, "stream": None}
        self.side = None
        self.ev_copied, self.ev_sampled = torch.cuda.Event(), torch.cuda.Event()
        self.ev_sampled.record()
        self.graph = None

    @staticmethod
    def _views(buf, like):
        out, o = [], 0
        for t in like:
            v = buf[o:o + t.numel()]
            out.append((v if t.dtype == torch.int32 else v.view(torch.float32)).view(t.shape))
            o += t.numel()
        return out

    def _body(self):
        self.dp.zero_grad()
        out = self.net(*self.inputs[:-1], sampling=self.samp)
        loss = self.loss_fn(out, self.inputs[-1])
        loss.backward()
        self.opt.step()
        return loss

    def capture(self, warmup=3):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._body()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.loss = self._body()
        self.graph = g

    def step(self, batch, next_batch, eager=False):
        from pointcloudlib_amd.networks.cls.pointnet2 import sampling_stream
        main = torch.cuda.current_stream()
        for d, s in zip(self.inputs, batch):
            d.copy_(s, non_blocking=True)
        main.wait_event(self.ev_sampled)
        self.cur.copy_(self.nxt, non_blocking=True)
        self.ev_copied.record(main)
        # the next batch's sampling FIRST (its producer stream waits for the main stream as enqueued so far -- the handle copy -- not for
        # the step that follows), then the step itself
        side, _ = sampling_stream(self.net, "own", batch[0].device)
        s = self.net.precompute_sampling(next_batch[0], stream="own")
        with torch.cuda.stream(side), torch.no_grad():
            torch._foreach_copy_(self.nxt_views, _flat(s))
            self.ev_sampled.record(side)
        if eager:
            self._body()
        else:
            self.graph.replay()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--windows", type=int, default=5)
    ap.add_argument("--host-delay-us", type=float, default=600.0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from pointcloudlib_amd import _lib, synth
    from pointcloudlib_amd.dp import FlatBucketDP
    from pointcloudlib_amd.networks.cls.pointnet2 import PointNet2_cls
    from pointcloudlib_amd.train_utils import make_sgd, soft_cross_entropy_loss
    _lib.lib()
    dev = torch.device("cuda", 0)
    B, N = 32, 1024
    torch.manual_seed(0)
    net = PointNet2_cls().to(dev).train()
    dp = FlatBucketDP(net)
    opt = make_sgd(net.parameters(), lr=0.02, momentum=0.9)
    batches = [(torch.from_numpy(synth.gauss_ball(B, N, 20242 + i)).to(dev), torch.from_numpy(synth.unit_normals(B, N, 20742 + i)).to(dev),
                torch.from_numpy(synth.labels(B, 40, 21142 + i)).to(dev)) for i in range(4)]
    loss_fn = lambda out, y: soft_cross_entropy_loss(out, y)

    # ---- eager reference: bench.py's step
    pending = {}
    def eager_step(i, delay):
        if delay:
            t_end = time.perf_counter() + delay * 1e-6
            while time.perf_counter() < t_end:
                pass
        x, f, y = batches[i % 4]
        dp.zero_grad()
        samp = pending.pop(i, None)
        pending[i + 1] = net.precompute_sampling(batches[(i + 1) % 4][0], stream="own")
        out = net(x, f, sampling=samp)
        loss_fn(out, y).backward()
        opt.step()

    it = [0]
    def run_eager(n, delay):
        for _ in range(n):
            eager_step(it[0], delay); it[0] += 1

    res = {"device": torch.cuda.get_device_name(0), "cpu_affinity": AFF, "steps": a.steps, "windows": a.windows}
    def timed(fn, delay, **kw):
        w = []
        for _ in range(a.windows):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            fn(a.steps, delay, **kw)
            th = time.perf_counter() - t0
            torch.cuda.synchronize(); w.append(((time.perf_counter() - t0) / a.steps * 1e3, th / a.steps * 1e3))
        return {"median_ms": round(statistics.median(x[0] for x in w), 4), "windows_ms": [round(x[0], 4) for x in w],
                "host_ms": [round(x[1], 4) for x in w]}
    import gc
    gc.collect(); gc.disable()
    run_eager(100, 0)
    # ---- 1. the eager step BEFORE anything is captured: the process has two streams, like bench.py's
    for d in (0.0, a.host_delay_us, 1500.0):
        res[f"eager_before_capture_delay{int(d)}"] = timed(run_eager, d)
    torch.cuda.synchronize()
    pending.clear()
    gs = GraphedStep(net, loss_fn, opt, dp, batches[0])
    err = None
    try:
        gs.capture()
    except Exception as e:                                   # noqa: BLE001 -- the result of the measurement is then "not capturable"
        err = f"{type(e).__name__}: {e}"[:500]
    res["capture_error"] = err
    jt = [0]
    def run_graph(n, delay, eager=False):
        for _ in range(n):
            if delay:
                t_end = time.perf_counter() + delay * 1e-6
                while time.perf_counter() < t_end:
                    pass
            gs.step(batches[jt[0] % 4], batches[(jt[0] + 1) % 4], eager=eager); jt[0] += 1
    # ---- 2. the replay (capture added a warm-up stream and torch's capture stream to the process), then the eager step AGAIN: HIP maps
    # streams onto a few hardware queues, and with four streams the sampling stream may land on the main stream's queue
    if err is None:
        run_graph(60, 0)
        for rep in range(2):
            for d in (0.0, a.host_delay_us, 1500.0):
                res[f"graph_delay{int(d)}_{rep}"] = timed(run_graph, d)
            res[f"eager_after_capture_{rep}"] = timed(run_eager, 0)
        res["static_buffers_eager_after_capture"] = timed(run_graph, 0, eager=True)
    print(json.dumps(res, indent=1))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
