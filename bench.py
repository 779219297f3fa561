#!/usr/bin/env python
"""bench.py -- point-clouds/sec, forward+backward(+SGD step), PointNet++ SSG cls, B=32 per GPU, N=1024.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement): whole-job clouds/s with inputs resident in
HBM, `roofline` for the dominant own kernel (HIP events on the launch stream inside the timed region) and
`cpu_baseline` (the CPU restatement of the reference semantics timed on the host cores; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pointcloudlib_amd.affinity import pin_to_gpu_node  # noqa: E402

# before torch / HIP start their threads: the rank's host thread stays on its GPU's socket (pointcloudlib_amd/affinity.py)
CPU_AFFINITY = pin_to_gpu_node(int(os.environ.get("LOCAL_RANK", "0")))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HOST_DELAY_US = float(os.environ.get("PCL_HOST_DELAY_US", "0"))
HOST_DELAY_AT = os.environ.get("PCL_HOST_DELAY_AT", "step")          # "step": at the head of the step; "bwd": between the loss and loss.backward()
PREFETCH_AT = os.environ.get("PCL_PREFETCH_AT", "fwd")     # where batch t+1's sampling is enqueued: beside step t's forward (default) or backward
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_PEAK_TFLOPS = 157.3     # fp32 vector == fp32-input MFMA peak


from pointcloudlib_amd.buildinfo import csrc_sha  # noqa: E402


def make_batches(B, N, n_batches, rank, dev, dist="gauss_ball"):
    from pointcloudlib_amd import synth
    cloud = getattr(synth, dist)
    out = []
    for i in range(n_batches):
        seed = 20242 + 1000 * rank + i
        out.append((torch.from_numpy(cloud(B, N, seed)).to(dev),
                    torch.from_numpy(synth.unit_normals(B, N, seed + 500)).to(dev),
                    torch.from_numpy(synth.labels(B, 40, seed + 900)).to(dev)))
    return out


def cpu_baseline(state, B, N, n_steps=10, n_warm=3):
    """The CPU restatement (oracle index ops, OpenMP over clouds + PyTorch-CPU fp32 dense ops) timed on the
    host cores on a bounded sample of the same workload: SURVEY 8d's protocol, `n_warm` warm-up steps then the MEDIAN of
    `n_steps` (>= 10) full fwd+bwd+SGD steps (~2 s each on the GPU box's host: ~30 s in all)."""
    import oracle
    from oracle.cpu_model import PointNet2ClsCPU
    from pointcloudlib_amd import synth
    from pointcloudlib_amd.train_utils import soft_cross_entropy_loss
    ncpu = os.cpu_count() or 1
    net = PointNet2ClsCPU(state, tie_stride=oracle.optimal_block(B)).train()
    net.use_dropout = True
    opt = torch.optim.SGD(net.parameters(), lr=0.02, momentum=0.9)
    x = torch.from_numpy(synth.gauss_ball(B, N, 20242))
    f = torch.from_numpy(synth.unit_normals(B, N, 20742))
    y = torch.from_numpy(synth.labels(B, 40, 21142))

    def step():
        opt.zero_grad()
        loss = soft_cross_entropy_loss(net(x, f), y)
        loss.backward()
        opt.step()

    # PyTorch-CPU does not scale to hundreds of threads on ops this small: pick the fastest of a few thread counts
    # with one untimed step each (the first of them after one plain step: these are the warm-up), then time with that count.
    step()
    best_t, cores = None, 1
    for c in sorted({min(ncpu, c) for c in (16, 32, 64, 128)}):
        torch.set_num_threads(c)
        t0 = time.perf_counter()
        step()
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best_t, cores = dt, c
    torch.set_num_threads(cores)
    for _ in range(max(0, n_warm - 1)):
        step()
    times = []
    for _ in range(n_steps):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    times.sort()
    n, dt = 1, times[len(times) // 2]                     # median step
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    return {"value": round(B * n / dt, 3), "unit": "point-clouds/s", "cores": cores, "kind": "port",
            "sample": f"median of {n_steps} fwd+bwd+SGD steps (min {times[0]:.2f} s, max {times[-1]:.2f} s) of PointNet++ SSG B={B} N={N} after {n_warm}+ warm-ups; best of 16/32/64/128 "
                      f"PyTorch threads = {cores} of {ncpu} logical CPUs (oracle FPS/ball-query: OpenMP over the {B} "
                      f"clouds; dense ops: PyTorch-CPU fp32; {model})",
            "label": "CPU restatement of reference semantics (Jittor not runnable)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--windows", type=int, default=5,
                    help="consecutive timed windows of --steps steps each (every one bracketed by barrier + synchronize, MAX over ranks); "
                         "the line reports the MEDIAN window and lists all of them")
    ap.add_argument("--rewarm", type=int, default=10, help="untimed product-path steps between the per-kernel profiling pre-pass and the timed region")
    ap.add_argument("--batch", type=int, default=32, help="clouds per GPU (BASELINE config 2: 32)")
    ap.add_argument("--npoints", type=int, default=1024, help="points per cloud (1024; 4096 is the north-star extra)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", dest="other_configs", action="store_false",
                    help="skip the short legs for BASELINE configs 1, 3, 4, 5 and config 2 on sphere_shell clouds / at N=4096 that follow the timed region (N=1 only; ~25 s)")
    ap.add_argument("--cpu-baselines", action="store_true",
                    help="every other_configs row also carries the CPU restatement of its own workload (tools/bench_models.py's leg: minutes; off by default so "
                         "that the default run stays inside a few minutes)")
    ap.add_argument("--roofline-kernel", default="auto")
    ap.add_argument("--no-prefetch-sampling", dest="prefetch_sampling", action="store_false",
                    help="run FPS/ball query inline at the head of each forward instead of one step ahead on a side stream")
    ap.add_argument("--profile-all", action="store_true", help="print a per-entry-point event-timed table to stderr")
    ap.add_argument("--dist", choices=["gauss_ball", "sphere_shell"], default="gauss_ball",
                    help="synthetic cloud distribution (SURVEY 8d): gauss_ball = the ModelNet40 loader's statistics (headline), "
                         "sphere_shell = never-saturating ball queries, ~80 %% padded duplicates")
    ap.add_argument("--dp-bucket-bytes", type=int, default=0, help="cut the gradient exchange (N>1) into buckets of this size; 0 = one all-reduce")
    ap.add_argument("--dp-overlap", action="store_true",
                    help="send gradient buckets from autograd hooks during backward (measured slower, see pointcloudlib_amd/dp.py)")
    ap.add_argument("--no-settle", dest="settle", action="store_false", help="skip the untimed clock-settling windows (profiler passes)")
    ap.add_argument("--dump-launch-order", default=None, help="write the (entry point, shape) sequence of one step as JSON")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = "RANK" in os.environ and "MASTER_PORT" in os.environ      # launched by torch.distributed.run
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)       # nccl == RCCL on ROCm

    from pointcloudlib_amd import _lib
    from pointcloudlib_amd.dp import FlatBucketDP
    from pointcloudlib_amd.networks.cls.pointnet2 import PointNet2_cls
    from pointcloudlib_amd.train_utils import loss_backward, make_sgd, soft_cross_entropy_loss
    _lib.lib()                                                # fail loudly when the extension is missing
    if os.environ.get("PCL_FPS_PRIO"):                        # lab switch (tools/ab.sh): issue priority of the FPS chain's waves
        _lib.lib().pcl_set_fps_tuning(0, int(os.environ["PCL_FPS_PRIO"]))
    if os.environ.get("PCL_MATRIX_FORM"):                     # lab switch (pcl_set_matrix_form): bit 0 resident forward, bit 1 staged GEMMs, min K << 8
        _lib.lib().pcl_set_matrix_form(int(os.environ["PCL_MATRIX_FORM"], 0))

    B, N = args.batch, args.npoints
    torch.manual_seed(0)
    net = PointNet2_cls().to(dev).train()
    state0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    dp = FlatBucketDP(net, bucket_bytes=args.dp_bucket_bytes, overlap=args.dp_overlap)
    opt = make_sgd(net.parameters(), lr=0.02, momentum=0.9)   # train_cls.py:374-377,404
    batches = make_batches(B, N, 4, rank, dev, args.dist)

    side = "own" if args.prefetch_sampling else None     # the network's private high-priority producer stream (pointnet2.sampling_stream)
    pending = {}

    def step(i):
        """One training step on batch i.  With --prefetch-sampling (default) the FPS/ball-query indices of batch
        i+1 are produced on a side stream while step i runs (input-pipeline style; enqueued ahead of its forward); every step still
        executes exactly one full set of index ops, and batch i's own set was produced during step i-1."""
        x, f, y = batches[i % len(batches)]
        def spin():                            # lab switch: is the host on the critical path?  (busy-wait, no GPU interaction)
            t_end = time.perf_counter() + HOST_DELAY_US * 1e-6
            while time.perf_counter() < t_end:
                pass
        if HOST_DELAY_US and HOST_DELAY_AT == "step":
            spin()
        dp.zero_grad()
        samp = pending.pop(i, None)
        if side is not None and PREFETCH_AT == "fwd":          # the next batch's sampling beside THIS step's forward (A/B: 1.900 vs 1.914 ms beside the backward)
            pending[i + 1] = net.precompute_sampling(batches[(i + 1) % len(batches)][0], stream=side)
        out = net(x, f, sampling=samp)
        if side is not None and PREFETCH_AT != "fwd":
            pending[i + 1] = net.precompute_sampling(batches[(i + 1) % len(batches)][0], stream=side)
        loss = soft_cross_entropy_loss(out, y)
        if HOST_DELAY_US and HOST_DELAY_AT == "bwd":
            spin()
        loss_backward(loss)
        dp.all_reduce()
        opt.step()
        return loss

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    it = 0
    for _ in range(args.warmup):
        step(it); it += 1
    # clocks, allocator pools and the sampling pipeline settle over the first ~50 ms; short runs (W=5, K=30) otherwise
    # swing by 10 % from run to run.  Untimed, like the W steps above.
    prev = None
    for _ in range(12 if args.settle else 0):                      # 10-step windows until two consecutive ones agree within 2 % (<= 120 steps)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            step(it); it += 1
        torch.cuda.synchronize(); cur = time.perf_counter() - t0
        stop = prev is not None and abs(cur - prev) <= 0.02 * cur
        if distributed:                      # every step holds a collective: all ranks must leave together
            flag = torch.tensor([1.0 if stop else 0.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            stop = bool(flag.item() > 0.5)
        if stop:
            break
        prev = cur
    # The per-kernel event timer brackets C-ABI calls; the product path makes ONE call per stack (csrc/stack.hip), so the
    # untimed profiling passes below (per-entry-point table, launch order, choice of the dominant kernel) run the same
    # kernels through the per-kernel entry points (mlp_hip.USE_STACK = False: bit-identical results, tests/test_mlp_hip.py),
    # and the TIMED region runs the product path with the events armed for the chosen kernel's launch tag.
    from pointcloudlib_amd.misc import mlp_hip
    from pointcloudlib_amd.misc import head as _head
    stack_default, head_default = mlp_hip.USE_STACK, _head.USE_STACK
    if args.profile_all:                                # every rank steps (collectives); rank 0 prints
        mlp_hip.per_kernel_path().__enter__()
        _lib.PROFILER = _lib.KernelTimer()
        for _ in range(3):
            step(it); it += 1
        torch.cuda.synchronize()
        summ = _lib.PROFILER.summary()
        _lib.PROFILER = None
        tot = sum(v["total_ms"] for v in summ.values()) / 3
        if rank == 0:
            print(f"--- own C-ABI calls, per step total {tot:.3f} ms", file=sys.stderr)
            for (name, tag), v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"]):
                gbs = v["algo_bytes"] / (v["avg_ms"] * 1e-3) / 1e9 if v["algo_bytes"] else 0
                tf = v["algo_flops"] / (v["avg_ms"] * 1e-3) / 1e12 if v["algo_flops"] else 0
                print(f"{name:28s} {tag:14s} n/step={v['launches'] / 3:4.1f} avg={v['avg_ms']:8.4f} ms  {gbs:8.1f} GB/s {tf:7.2f} TF",
                      file=sys.stderr)
        mlp_hip.USE_STACK = stack_default; _head.USE_STACK = head_default
    if args.dump_launch_order:
        mlp_hip.per_kernel_path().__enter__()
        _lib.PROFILER = _lib.KernelTimer()
        step(it); it += 1
        torch.cuda.synchronize()
        if rank == 0:
            json.dump({"step_launch_order": _lib.PROFILER.order}, open(args.dump_launch_order, "w"))
        _lib.PROFILER = None
        mlp_hip.USE_STACK = stack_default; _head.USE_STACK = head_default
    # pick the dominant own kernel (untimed 2-step pre-pass with every entry point bracketed by events)
    target = args.roofline_kernel
    target_algo = None
    if target == "auto":
        mlp_hip.per_kernel_path().__enter__()
        _lib.PROFILER = _lib.KernelTimer()
        step(it); it += 1
        step(it); it += 1
        torch.cuda.synchronize()
        summ = _lib.PROFILER.summary()
        _lib.PROFILER = None
        mlp_hip.USE_STACK = stack_default; _head.USE_STACK = head_default
        torch.cuda.synchronize()
        # dominant = the C-ABI entry point with the largest total time, then its most expensive launch shape
        # (FPS is excluded: it is a latency-bound chain that runs on the side stream beside the GEMMs)
        by_name = {}
        for (name, tag), v in summ.items():
            if name != "pcl_fps_f32":
                by_name[name] = by_name.get(name, 0.0) + v["total_ms"]
        target = None
        if by_name:
            top = max(by_name, key=by_name.get)
            target = max(((k, v) for k, v in summ.items() if k[0] == top), key=lambda kv: kv[1]["total_ms"])[0]
            target_algo = (summ[target]["algo_bytes"], summ[target]["algo_flops"])
    elif target == "none":
        target = None
    else:
        target = (target, None)
    launch = "eager"
    timer = None
    if target:
        # HIP events around the dominant kernel's dominant launch shape only, a few launches of the timed region
        # (every timing event is a marker packet on the stream; bracketing everything would perturb `value`)
        in_stack = (stack_default and target_algo is not None and target[1] is not None
                    and target[0] in _lib.KERNEL_TIMED)        # a GEMM-family kernel that the product path launches from a stack call
        if in_stack:
            timer = _lib.KernelTimer(max_records=64, inner=(target[0], target[1], target_algo[0], target_algo[1]))
        else:
            timer = _lib.KernelTimer([target[0]], tags=None if target[1] is None else [target[1]], max_records=64)

    import gc
    lab_no_gc = bool(os.environ.get("PCL_BENCH_NO_GC"))                   # lab switches (tools/first_window.sh): what makes the FIRST window slow?
    lab_idle_ms = float(os.environ.get("PCL_BENCH_IDLE_MS", "0"))         # host sleep (GPU idle) in front of window 2
    if not lab_no_gc:
        gc.collect()
        gc.disable()             # no collector pauses inside the timed region (a gen-2 pass over the autograd objects is ~5 ms: a window of 5.3 ms per step, measured)
    # Round 6, root cause of round 5's driver line (2.135 ms where the same run's N = 4096 row took 2.187): the GPU IDLES through the
    # profiling pre-pass's summary and the full gc.collect() above (tens of ms), and the first ~40 ms of work after an idle period run 4-5 %
    # slow (14 of 14 runs: first window 1.93-1.97 ms, the four behind it 1.86-1.87; without the collect the first window is 1.863; a 100 / 400 ms
    # host sleep in front of ANY window makes that window 1.93 / 1.95 -- gpurun_out/r06a, r06b, tools/first_window.sh).  So the untimed
    # re-warm steps come AFTER the collect, and run until two consecutive 10-step windows agree within 1 % (at most 100 steps), like the
    # settling pass in front of the pre-pass.
    prev = None
    for k in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10 if k or args.rewarm <= 10 else args.rewarm):
            step(it); it += 1
        torch.cuda.synchronize(); cur = time.perf_counter() - t0
        stop = k >= 1 and prev is not None and abs(cur - prev) <= 0.01 * cur
        if distributed:
            flag = torch.tensor([1.0 if stop else 0.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            stop = bool(flag.item() > 0.5)
        if stop or not args.settle and k >= 1:
            break
        prev = cur
    # The timed region: `--windows` consecutive windows of EXACTLY `--steps` steps, each bracketed by barrier + synchronize on both
    # sides.  ms_per_step / value come from the MEDIAN window (MAX over ranks per window); every window is listed in the line.  One
    # window of 20 steps is 40 ms -- the size of one host hiccup or one clock ramp (round 5's driver line: 2.135 ms from a single window
    # whose neighbours ran 1.9) -- so a single window is not a measurement of the kernels; the same estimator serves `other_configs`.
    # The dominant kernel's event pair is armed in every OTHER window (1, 3, ...): `windows_timer_armed` in the line shows whether
    # the in-stack timer moves a window (it is two events filled from the kernel's own dispatch packet, no marker packets).
    host_trace = [] if os.environ.get("PCL_HOST_TRACE") else None      # lab switch: when does the HOST leave each step?
    win_dt, win_host, win_armed = [], [], []
    if timer is not None:
        timer.max_records = None
    for w in range(max(1, args.windows)):
        # (armed in windows 1, 3, ...: the event pair costs ~8 us per step -- 12 of 12 runs on one box: armed windows 1.809-1.815 ms, the
        #  unarmed ones between them 1.800-1.807 -- and the product path does not carry it, so the MAJORITY of the windows, and with it the
        #  median, is unarmed; a single window is armed, or the line would have no live kernel time)
        armed = timer is not None and (w % 2 == 1 or max(1, args.windows) == 1)
        _lib.PROFILER = timer if armed else None
        win_armed.append(armed)
        if lab_idle_ms and w == 2:
            torch.cuda.synchronize(); time.sleep(lab_idle_ms * 1e-3)
        fence()
        t0 = time.perf_counter()
        for j in range(args.steps):
            step(it)
            it += 1
            if host_trace is not None:
                host_trace.append(time.perf_counter() - t0)
        th = time.perf_counter() - t0
        fence()
        win_dt.append(time.perf_counter() - t0)
        win_host.append(th)
        if host_trace is not None and rank == 0:
            print(f"window {w}: host left step j at [ms]: " + " ".join(f"{t * 1e3:.2f}" for t in host_trace) + f" | fence at {win_dt[-1] * 1e3:.2f}", file=sys.stderr)
            host_trace.clear()
    _lib.PROFILER = None
    # host side of a step (untimed extras): the time Python needs to ENQUEUE one step with the stream empty behind it (4
    # steps back to back without a sync; the launch queue is deeper than that), and the number of own C-ABI launches
    host_ms = own_launches = None
    if True:
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        for _ in range(4):
            step(it); it += 1
        host_ms = (time.perf_counter() - h0) / 4 * 1e3
        fence()
        counter = _lib.KernelTimer(names=[])            # matches nothing: counts through .order without recording events
        _lib.PROFILER = counter
        step(it); it += 1
        _lib.PROFILER = None
        own_launches = counter.calls
        fence()
    gc.enable()
    if distributed:
        t = torch.tensor(win_dt, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        win_dt = [float(v) for v in t.tolist()]
    import statistics
    dt = statistics.median(win_dt)

    if rank == 0:
        roofline = None
        if timer is not None:
            summ = timer.summary()
            key = target if target[1] is not None and target in summ else max(summ, key=lambda k: summ[k]["total_ms"])
            r = summ[key]
            # which roof bounds this launch: arithmetic intensity against the machine balance (157.3 TF / 8 TB/s)
            ai = r["algo_flops"] / max(1.0, r["algo_bytes"])
            bound = "mfma" if ai > FP32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9) else "hbm"
            if bound == "hbm":
                ach = r["algo_bytes"] / (r["avg_ms"] * 1e-3) / 1e9
                peak, unit = HBM_PEAK_GBS, "GB/s"
            else:
                ach = r["algo_flops"] / (r["avg_ms"] * 1e-3) / 1e12
                peak, unit = FP32_PEAK_TFLOPS, "TFLOP/s"
            # PMC HBM bytes per launch come from separate rocprofv3 --pmc passes of this same command (tools/pmc_traffic.py ->
            # profiles/rNN_traffic.json, the newest one).  They are only valid for the kernel sources they were measured on: the file records
            # the git blob hashes of csrc/*.hip + common.h, and a mismatch (or another workload) prints null.
            traffic, traffic_src = None, None
            try:
                from pointcloudlib_amd.buildinfo import traffic_profile
                tj = traffic_profile()
                if tj is not None and (B, N, args.dist) == (32, 1024, "gauss_ball"):
                    traffic = tj["per_launch_hbm_bytes"].get(f"{key[0]}:{key[1]}")
                # (the pass writes under gpurun_out/ on the GPU box; the committed copy of the same file is under profiles/)
                traffic_src = ("profiles/" + os.path.basename(tj["source"])) if traffic is not None else None
            except Exception:
                pass
            roofline = {"kernel": key[0], "shape": key[1], "bound": bound, "achieved": round(ach, 3), "peak": peak, "unit": unit,
                        "frac": round(ach / peak, 5), "traffic": traffic, "traffic_source": traffic_src,
                        "avg_launch_ms": round(r["avg_ms"], 5),
                        "launches": r["launches"], "algo_bytes_per_launch": r["algo_bytes"],
                        "algo_flops_per_launch": r["algo_flops"]}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(state0, B, N)
        others = None
        if world == 1 and not distributed and args.other_configs and (B, N, args.dist) == (32, 1024, "gauss_ball"):
            # after the headline's timed region and its CPU leg: the other BASELINE workloads, 20 train steps each on this GPU with the
            # roofline of their own dominant kernel (tools/bench_models.py; builder-run copies with kernel-stat CSVs under profiles/)
            # Nothing in this leg may cost the headline its line: every config is isolated inside other_configs(), and a failure of the
            # leg itself (import, allocator) is reported in place of the rows.
            try:
                del batches[:]
                pending.clear()
                torch.cuda.empty_cache()
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_models
                others = bench_models.other_configs(steps=20, windows=max(1, args.windows), cpu_baselines=args.cpu_baselines)
            except Exception as e:                                  # noqa: BLE001 -- reported in the line
                others = [{"key": "other_configs", "error": f"{type(e).__name__}: {e}"[:400]}]
        value = world * B * args.steps / dt
        line = {
            "metric": f"point-clouds/sec fwd+bwd, PointNet++ SSG B={B} N={N}", "value": round(value, 2),
            "unit": "point-clouds/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "timing": f"median of {len(win_dt)} consecutive windows of {args.steps} steps, each bracketed by barrier + synchronize (MAX over ranks per window)",
            "windows_ms_per_step": [round(v / args.steps * 1e3, 4) for v in win_dt],
            "windows_host_enqueue_ms_per_step": [round(v / args.steps * 1e3, 4) for v in win_host],
            "windows_timer_armed": win_armed,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"PointNet++ SSG cls train step (fwd+bwd+SGD), B={B}/GPU, N={N} xyz+normal, "
                                   f"{args.dist} clouds (BASELINE configs[1])",
                       "global_batch": world * B, "n_points": N, "parallelism": f"dp{world}",
                       "sampling": ("indices of batch t+1 on a side stream during step t (enqueued before its " + ("forward" if PREFETCH_AT == "fwd" else "backward") + ")") if args.prefetch_sampling
                       else "inline",
                       "launch": launch, "grad_bucket_bytes": dp.bucket_nbytes, "grad_overlap": bool(dp.overlap and dp.active), "cpu_affinity": CPU_AFFINITY,
                       "world_size": dist.get_world_size() if distributed else 1, "backend": dist.get_backend() if distributed else None,
                       "rccl": ".".join(map(str, torch.cuda.nccl.version())) if distributed else None,
                       "sync_bn": bool(getattr(dp, "sync_bn", False)),
                       "entry_points": "per-stack (pcl_mlp_stack_*_f32)" if stack_default else "per-kernel"},
            "own_launches_per_step": own_launches, "host_enqueue_ms": None if host_ms is None else round(host_ms, 3),
            "roofline": roofline, "cpu_baseline": cpu, "other_configs": others,
        }
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
