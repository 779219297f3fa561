"""One training step as a HIP graph, with the next batch's sampling beside it.

The PointNet++ step is ~170 kernel launches; enqueueing them from Python costs about as much host time as the GPU needs
to run them.  Everything that runs on the main stream -- zero_grad, forward, loss, backward, (single-process) optimizer --
is captured once into a HIP graph and replayed with ONE launch per step.  The index-producing ops (FPS, ball query, group
offsets) of the NEXT batch stay eager on a high-priority side stream, so that their latency-bound chain overlaps the
graph: a hipGraph with that second branch captured inside replayed 60 % slower than the eager step (measured), a
single-stream graph replays at the sum of its kernel times.

Data flow per step i:   main:  S_cur <- S_next (one 5 MB copy) ; replay(graph reading x, f, y, S_cur)
                        side:  wait(copy done) ; S_next <- sample(batch i+1)
``S_cur`` / ``S_next`` are flat static buffers holding every sampling tensor (centres, neighbour lists, counts, offsets).
With ``eager=True`` a step runs the same data flow through the normal Python path (used for the event-timed steps).
"""
import torch


def _flat(sampling):
    return [t for new_xyz, idxs in sampling["levels"] for t in [new_xyz] + [u for ic in idxs if ic is not None for u in ic]
            if t is not None]


class GraphedStep:
    def __init__(self, net, loss_fn, optimizer, dp, sample_xyz, example_batch, side_stream, capture_optimizer=True):
        self.net, self.loss_fn, self.opt, self.dp = net, loss_fn, optimizer, dp
        dp.overlap = False                                    # no collectives from autograd hooks inside a capture
        self.side = side_stream
        if getattr(dp, "active", False) and getattr(dp, "world", 1) > 1 and capture_optimizer:
            # a captured opt.step() would run before any gradient exchange: the replicas would silently diverge
            raise ValueError("GraphedStep: capture_optimizer=True cannot be combined with an active multi-rank FlatBucketDP; "
                             "pass capture_optimizer=False (the all-reduce and the update then run eagerly after the replay)")
        self.capture_optimizer = capture_optimizer
        self.inputs = [torch.empty_like(t) for t in example_batch]
        for d, s in zip(self.inputs, example_batch):
            d.copy_(s)
        with torch.no_grad():
            s0 = net.precompute_sampling(sample_xyz(example_batch))
        flat = _flat(s0)
        assert all(t.element_size() == 4 for t in flat)
        n = sum(t.numel() for t in flat)
        dev = flat[0].device
        self.cur = torch.empty(n, dtype=torch.int32, device=dev)
        self.nxt = torch.empty(n, dtype=torch.int32, device=dev)
        self.cur_views = self._views(self.cur, flat)
        self.nxt_views = self._views(self.nxt, flat)
        self._pack(s0, self.nxt_views)
        self.cur.copy_(self.nxt)
        # the sampling handle the captured forward reads: views into the static buffer
        it = iter(self.cur_views)
        levels = []
        for new_xyz, idxs in s0["levels"]:
            nx = next(it) if new_xyz is not None else None
            levels.append((nx, [None if ic is None else tuple(next(it) if u is not None else None for u in ic) for ic in idxs]))
        self.samp = {"levels": levels, "event": None, "stream": None}
        self.sample_xyz = sample_xyz
        self.ev_copied = torch.cuda.Event()
        self.ev_sampled = torch.cuda.Event()
        self.ev_sampled.record()
        self.graph = None
        self.loss = None
        self._dirty = False

    @staticmethod
    def _views(buf, like):
        out, o = [], 0
        for t in like:
            v = buf[o:o + t.numel()]
            out.append((v if t.dtype == torch.int32 else v.view(torch.float32)).view(t.shape))
            o += t.numel()
        return out

    @staticmethod
    def _pack(sampling, views):
        torch._foreach_copy_(views, _flat(sampling))

    def _body(self):
        self.dp.zero_grad()
        out = self.net(*self.inputs[:-1], sampling=self.samp)
        loss = self.loss_fn(out, self.inputs[-1])
        loss.backward()
        if self.capture_optimizer:
            self.opt.step()
        return loss

    def capture(self, warmup=3):
        """Warm up on a side stream (allocator / lazy initialisation), then capture the main-stream part of a step."""
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._body()
                if not self.capture_optimizer:
                    self.opt.step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.loss = self._body()
        self.graph = g
        self.static_grads = [(p, p.grad) for p in self.net.parameters() if p.grad is not None]
        self._dirty = False

    def step(self, batch, next_batch, eager=False):
        main = torch.cuda.current_stream()
        for d, s in zip(self.inputs, batch):
            d.copy_(s, non_blocking=True)
        main.wait_event(self.ev_sampled)                      # S_next holds this batch's sampling
        self.cur.copy_(self.nxt, non_blocking=True)
        self.ev_copied.record(main)
        if eager or self.graph is None:
            loss = self._body()
            self._dirty = self.graph is not None          # param.grad now point at this step's fresh tensors
        else:
            if self._dirty:                               # back to the tensors the graph writes
                for p, g in self.static_grads:
                    p.grad = g
                self._dirty = False
            self.graph.replay()
            loss = self.loss
        # the next batch's sampling, beside the work just enqueued
        self.side.wait_event(self.ev_copied)
        with torch.cuda.stream(self.side), torch.no_grad():
            s = self.net.precompute_sampling(self.sample_xyz(next_batch), stream=self.side)
            self._pack(s, self.nxt_views)
            self.ev_sampled.record(self.side)
        if not self.capture_optimizer:
            self.dp.all_reduce_into_grads()
            self.opt.step()
        return loss
