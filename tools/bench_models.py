"""Throughput of the other BASELINE configs on one MI355X (train step: fwd + bwd + SGD, synthetic data).
    python tools/bench_models.py [--steps 10] [--out profiles/r01_other_configs.json]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudlib_amd import synth
from pointcloudlib_amd.train_utils import loss_backward, make_sgd, seg_cross_entropy_loss, soft_cross_entropy_loss
from pointcloudlib_amd.affinity import pin_to_gpu_node
pin_to_gpu_node(0)


FP32_PEAK_TFLOPS, HBM_PEAK_GBS = 157.3, 8000.0        # MI355X_MICROARCH.md: dense fp32 MFMA, HBM3E
ONLY = None                                            # --only: substring filter on the config names
TABLE = 0                                              # --table N: print the N most expensive (entry point, shape) rows of each config
ORDER_OUT = None                                       # --dump-launch-order FILE: the (entry point, shape, kernel) sequence of one step
QUIET = False                                          # bench.py's other_configs leg: no per-config lines on stdout


def roofline_of(step):
    """Dominant own kernel of the step (every C-ABI entry point event-timed for two untimed steps; GEMM-family kernels report
    their own begin / end timestamps) and its roofline, as bench.py does for the headline config."""
    from pointcloudlib_amd import _lib
    from pointcloudlib_amd.misc.mlp_hip import per_kernel_path
    with per_kernel_path():          # the event timer brackets C-ABI calls: one call per kernel for this untimed pass (same kernels)
        _lib.PROFILER = _lib.KernelTimer()
        step(); step()
        torch.cuda.synchronize()
        summ = _lib.PROFILER.summary()
        _lib.PROFILER = None
    if ORDER_OUT:
        _lib.PROFILER = _lib.KernelTimer()
        with per_kernel_path():
            step()
        torch.cuda.synchronize()
        json.dump({"step_launch_order": _lib.PROFILER.order}, open(ORDER_OUT, "w"))
        _lib.PROFILER = None
    if TABLE:
        for (n, t), v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])[:TABLE]:
            gbs = v["algo_bytes"] / (v["avg_ms"] * 1e-3) / 1e9 if v["algo_bytes"] else 0
            tf = v["algo_flops"] / (v["avg_ms"] * 1e-3) / 1e12 if v["algo_flops"] else 0
            print(f"    {n:32s} {t:16s} n/step={v['launches'] / 2:4.1f} avg={v['avg_ms'] * 1e3:8.1f} us  {gbs:8.1f} GB/s {tf:7.2f} TF", file=sys.stderr)
    by_name = {}
    # (entry points that state their algorithmic bytes / flops; the bookkeeping launches -- BatchNorm finalize, constants -- have no roofline
    # to be priced against and, summed over a small network's many layers, would otherwise be named "dominant": config 1)
    priced = any(v["algo_bytes"] or v["algo_flops"] for (n, t), v in summ.items() if n != "pcl_fps_f32")
    for (n, t), v in summ.items():
        if n != "pcl_fps_f32" and (not priced or v["algo_bytes"] or v["algo_flops"]):      # FPS: latency-bound chain, runs beside the GEMMs where sampling is prefetched
            by_name[n] = by_name.get(n, 0.0) + v["total_ms"]
    if not by_name:
        return None
    top = max(by_name, key=by_name.get)
    key, r = max(((k, v) for k, v in summ.items() if k[0] == top), key=lambda kv: kv[1]["total_ms"])
    own_ms = sum(v["total_ms"] for v in summ.values()) / 2
    ai = r["algo_flops"] / max(1.0, r["algo_bytes"])
    if ai > FP32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9):
        bound, ach, peak, unit = "mfma", r["algo_flops"] / (r["avg_ms"] * 1e-3) / 1e12, FP32_PEAK_TFLOPS, "TFLOP/s"
    else:
        bound, ach, peak, unit = "hbm", r["algo_bytes"] / (r["avg_ms"] * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
    return {"kernel": key[0], "shape": key[1], "bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit,
            "frac": round(ach / peak, 4), "avg_launch_ms": round(r["avg_ms"], 5), "launches_per_step": r["launches"] / 2,
            "entry_point_share_of_own_kernel_time": round(by_name[top] / 2 / own_ms, 3), "own_kernel_ms_per_step": round(own_ms, 3),
            "traffic": None,
            **({"note": "exact-rounding distances run on the fp32 VALU, not MFMA: ceiling 64.4 Tops/s measured (tools/ubench/pkrate.hip)",
                "valu_peak_tops": 64.4, "frac_valu": round(ach / 64.4, 4)} if key[0] == "pcl_knn_f32" else {})}


CPU_BASELINE = False                                   # --cpu-baseline: also time the CPU restatement of the config
CPU_JOBS = []                                          # (result dict, kind, CPU state dict), run after the GPU configs


def cpu_baseline_of(kind, state, n_warm=1, n_steps=5):
    """the CPU restatement of the reference semantics (oracle/: index ops in C with OpenMP over the clouds, dense ops
    PyTorch-CPU fp32) on the same synthetic batch, fwd + bwd + SGD, median of `n_steps` after `n_warm` (a bounded sample:
    these steps take seconds).  Test infrastructure used as the CPU leg only -- nothing of it runs in the GPU path."""
    import statistics
    import numpy as np
    import oracle
    from oracle.cpu_dgcnn import DGCNNCPU
    from oracle.cpu_model import PointNet2ClsCPU
    from oracle.cpu_partseg import PointNet2PartSegCPU
    from oracle.cpu_pointconv import PointConvClsCPU
    lab = lambda B: torch.from_numpy(synth.labels(B, 40, 1))
    if kind == "cfg1":
        from oracle.cpu_pointnet import PointNetClsCPU
        B, N = 8, 1024
        net = PointNetClsCPU(state).train()
        x, y = torch.from_numpy(synth.gauss_ball(B, N, 20241)).transpose(1, 2).contiguous(), lab(B)
        fwd = lambda: soft_cross_entropy_loss(net(x), y)
    elif kind == "cfg2_n4096":
        B, N = 32, 4096
        net = PointNet2ClsCPU(state, tie_stride=oracle.optimal_block(B)).train()
        x, f, y = torch.from_numpy(synth.gauss_ball(B, N, 20242)), torch.from_numpy(synth.unit_normals(B, N, 7)), lab(B)
        fwd = lambda: soft_cross_entropy_loss(net(x, f), y)
    elif kind == "cfg3":
        B, N = 32, 1024
        net = DGCNNCPU(state, 20)
        x, y = torch.from_numpy(synth.gauss_ball(B, N, 20242)).transpose(1, 2).contiguous(), lab(B)
        fwd = lambda: soft_cross_entropy_loss(net(x), y)
    elif kind in ("cfg4", "cfg4_msg"):
        B, N = 16, 2048
        net = PointNet2PartSegCPU(state, PointNet2PartSegCPU.MSG if kind == "cfg4_msg" else PointNet2PartSegCPU.SSG, tie_stride=oracle.optimal_block(B))
        x = torch.from_numpy(synth.gauss_ball(B, N, 20244))
        oh = torch.zeros(B, 16); oh[torch.arange(B), torch.arange(B) % 16] = 1
        seg = torch.from_numpy(np.random.default_rng(5).integers(0, 50, (B, N)))
        fwd = lambda: torch.nn.functional.cross_entropy(net(x, x, oh), seg)
    elif kind == "cfg5":
        B, N = 32, 1024
        net = PointConvClsCPU(state)
        x, y = torch.from_numpy(synth.gauss_ball(B, N, 20242)).transpose(1, 2).contiguous(), lab(B)
        rng = np.random.default_rng(9)
        start = [rng.integers(0, N, B).astype(np.int32), rng.integers(0, 512, B).astype(np.int32)]
        fwd = lambda: soft_cross_entropy_loss(net(x, start), y)
    else:
        return None
    opt = torch.optim.SGD(net.parameters(), lr=0.02, momentum=0.9)
    threads = min(32, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    times = []
    for i in range(n_warm + n_steps):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        fwd().backward()
        opt.step()
        if i >= n_warm:
            times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    return {"value": round(B / med, 3), "unit": "point-clouds/s", "cores": threads, "kind": "port",
            "sample": f"median of {n_steps} fwd+bwd+SGD steps (min {min(times):.2f} s, max {max(times):.2f} s) after {n_warm} warm-up, "
                      f"B={B} N={N}; CPU restatement of reference semantics (oracle/, Jittor not runnable), {threads} PyTorch threads"}


def _selected(name):
    """--only SUBSTRING, or --only 'NAME$' for an exact match (a config whose name is a prefix of its prefetch variant's)"""
    if not ONLY:
        return True
    return name == ONLY[:-1] if ONLY.endswith("$") else ONLY in name


WINDOWS = []                                           # the windows of the last _timed() call, seconds per step
N_WINDOWS = 3                                          # --windows: consecutive timed windows per config (the median is reported)


def _timed(step, steps, windows=1):
    """seconds per step: the MEDIAN of `windows` consecutive windows of `steps` steps, each bracketed by a synchronize -- bench.py's
    estimator for the headline (one window unless a caller asks for more; every window is kept in WINDOWS for the row)"""
    import statistics
    del WINDOWS[:]
    for _ in range(windows):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(); WINDOWS.append((time.perf_counter() - t0) / steps)
    return statistics.median(WINDOWS)


def run(name, make, inputs, loss_fn, steps, warmup=3, cpu_kind=None, windows=None):
    if not _selected(name):
        return None
    torch.manual_seed(0)
    net = make().cuda().train()
    opt = make_sgd(net.parameters(), lr=0.02, momentum=0.9)
    def step():
        opt.zero_grad(set_to_none=True)
        loss_backward(loss_fn(net(*inputs)))
        opt.step()
    for _ in range(warmup):
        step()
    dt = _timed(step, steps, windows or N_WINDOWS)
    B = inputs[0].shape[0]
    r = {"config": name, "ms_per_step": round(dt * 1e3, 3), "clouds_per_s": round(B / dt, 1), "batch": B,
         "windows_ms_per_step": [round(w * 1e3, 3) for w in WINDOWS],
         "params": sum(p.numel() for p in net.parameters()), "roofline": roofline_of(step)}
    if CPU_BASELINE and cpu_kind:
        # (timed after ALL GPU configs: the CPU legs spin up 32 OpenMP / PyTorch threads that would compete with the Python
        # launch thread of the host-bound configs that follow)
        CPU_JOBS.append((r, cpu_kind, {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}))
    if not QUIET:
        print(json.dumps(r), flush=True)
    return r


def run_prefetch(name, make, inputs, loss_fn, steps, warmup=3):
    if not _selected(name):
        return None
    return _run_prefetch(name, make, inputs, loss_fn, steps, warmup)


def _run_prefetch(name, make, inputs, loss_fn, steps, warmup=120, windows=None):    # (two streams: the allocator pools settle over tens of steps)
    """Same, with the encoder's FPS / ball query of the next batch issued on a side stream beside the backward pass
    (networks with ``precompute_sampling``; the input is the same tensor every step, the work is not)."""
    torch.manual_seed(0)
    net = make().cuda().train()
    opt = make_sgd(net.parameters(), lr=0.02, momentum=0.9)
    side = "own"                      # the network's private producer stream
    pending = [None]
    def step():
        opt.zero_grad(set_to_none=True)
        cur = pending[0]
        pending[0] = net.precompute_sampling(inputs[0], stream=side)      # enqueued ahead of the forward (bench.py: 1.900 vs 1.914 ms behind it)
        out = net(*inputs, sampling=cur)
        loss_backward(loss_fn(out))
        opt.step()
    for _ in range(warmup):
        step()
    dt = _timed(step, steps, windows or N_WINDOWS)
    B = inputs[0].shape[0]
    r = {"config": name, "ms_per_step": round(dt * 1e3, 3), "clouds_per_s": round(B / dt, 1), "batch": B,
         "windows_ms_per_step": [round(w * 1e3, 3) for w in WINDOWS],
         "params": sum(p.numel() for p in net.parameters())}
    if not QUIET:
        print(json.dumps(r), flush=True)
    return r


def traffic_of(key, roofline):
    """PMC HBM bytes per launch of the row's dominant kernel from the newest profiles/rNN_traffic_<key>.json (two separate rocprofv3 --pmc
    passes of `tools/bench_models.py --only ...`, tools/prof_cfg.sh) -- only while the kernel sources still hash to what was measured."""
    from pointcloudlib_amd.buildinfo import traffic_profile
    try:
        tj = traffic_profile("_" + key)
        if tj is not None:
            return tj["per_launch_hbm_bytes"].get(f"{roofline['kernel']}:{roofline['shape']}"), tj["source"]
    except Exception:
        pass
    return None, None


def other_configs(steps=20, keys=("cfg1", "cfg2_sphere_shell", "cfg2_n4096", "cfg3", "cfg4", "cfg5"), windows=5, cpu_baselines=False):
    """bench.py's `other_configs` leg: the BASELINE workloads besides the headline (config 1; config 2 on the never-saturating
    sphere_shell clouds of SURVEY 8d and at N = 4096; configs 3, 4 (MSG), 5), timed with the headline's estimator -- the MEDIAN of
    `windows` consecutive windows of `steps` train steps, every window listed -- each with the roofline of its dominant kernel;
    networks with a sampling front end are timed both inline and with the headline's one-batch-ahead protocol.  With
    `cpu_baselines` every row also carries the CPU restatement of its workload (tens of seconds each: off by default)."""
    global QUIET, CPU_BASELINE
    from pointcloudlib_amd.networks.cls.pointnet import PointNet
    from pointcloudlib_amd.networks.cls.pointnet2 import PointNet2_cls
    from pointcloudlib_amd.networks.cls.dgcnn import DGCNN
    from pointcloudlib_amd.networks.cls.pointconv import PointConvDensityClsSsg
    from pointcloudlib_amd.networks.seg.pointnet2_partseg import PointNetMSG
    dev = "cuda"
    cloud = lambda B, N, seed: torch.from_numpy(synth.gauss_ball(B, N, seed)).to(dev)
    y32 = torch.from_numpy(synth.labels(32, 40, 1)).to(dev)
    ce = lambda o: soft_cross_entropy_loss(o, y32)
    timing = f"median of {windows} consecutive windows of `steps` steps after 10+ warm-up steps (bench.py's estimator)"
    def one(key):
        pre = None
        if key == "cfg1":
            y8 = torch.from_numpy(synth.labels(8, 40, 1)).to(dev)
            r = run("PointNet cls B=8 N=1024 (BASELINE configs[0] on the GPU path)", PointNet, (cloud(8, 1024, 20241).transpose(1, 2).contiguous(),),
                    lambda o: soft_cross_entropy_loss(o, y8), steps, warmup=10, windows=windows, cpu_kind="cfg1")
        elif key == "cfg2_sphere_shell":
            x = torch.from_numpy(synth.sphere_shell(32, 1024, 20242)).to(dev)
            inp = (x, torch.from_numpy(synth.unit_normals(32, 1024, 7)).to(dev))
            r = run("PointNet++ SSG cls B=32 N=1024, sphere_shell clouds (SURVEY 8d: ball queries never saturate)", PointNet2_cls, inp, ce, steps, warmup=10, windows=windows)
            pre = (PointNet2_cls, inp, ce)
        elif key == "cfg2_n4096":
            x = cloud(32, 4096, 20242)
            inp = (x, torch.from_numpy(synth.unit_normals(32, 4096, 7)).to(dev))
            r = run("PointNet++ SSG cls B=32 N=4096 (north_star's second cloud size)", PointNet2_cls, inp, ce, steps, warmup=10, windows=windows, cpu_kind="cfg2_n4096")
            pre = (PointNet2_cls, inp, ce)
        elif key == "cfg3":
            r = run("DGCNN cls B=32 N=1024 k=20 (BASELINE configs[2])", DGCNN, (cloud(32, 1024, 20242).transpose(1, 2).contiguous(),), ce, steps, warmup=10, windows=windows, cpu_kind="cfg3")
        elif key == "cfg4":
            xs = cloud(16, 2048, 20244)
            oh = torch.zeros(16, 16, device=dev); oh[torch.arange(16), torch.arange(16) % 16] = 1
            seg = torch.randint(0, 50, (16, 2048), device=dev)
            lossf = lambda o: seg_cross_entropy_loss(o, seg)
            r = run("PointNet++ MSG part-seg B=16 N=2048 (BASELINE configs[3])", PointNetMSG, (xs, xs, oh), lossf, steps, warmup=10, windows=windows, cpu_kind="cfg4_msg")
            pre = (PointNetMSG, (xs, xs, oh), lossf)
        elif key == "cfg5":
            inp = (cloud(32, 1024, 20242).transpose(1, 2).contiguous(),)
            r = run("PointConv cls B=32 N=1024 (BASELINE configs[4])", PointConvDensityClsSsg, inp, ce, steps, warmup=10, windows=windows, cpu_kind="cfg5")
            pre = (PointConvDensityClsSsg, inp, ce)
        else:
            return None
        row = {"key": key, "workload": r["config"] + ", train step fwd+bwd+SGD", "ms_per_step": r["ms_per_step"],
               "value": r["clouds_per_s"], "unit": "point-clouds/s", "steps": steps, "timing": timing,
               "windows_ms_per_step": r["windows_ms_per_step"], "sampling": "inline", "roofline": r["roofline"], "_r": r}
        if pre is not None:
            # the headline's protocol: the coordinate-only work of batch t+1 (FPS, ball query / k-NN groups, kernel densities) on the
            # network's side stream during step t; the inline figure stays beside it
            rp = _run_prefetch(r["config"], pre[0], pre[1], pre[2], steps, warmup=40, windows=windows)
            row.update({"ms_per_step_inline": r["ms_per_step"], "windows_ms_per_step_inline": r["windows_ms_per_step"],
                        "ms_per_step": rp["ms_per_step"], "windows_ms_per_step": rp["windows_ms_per_step"], "value": rp["clouds_per_s"],
                        "sampling": "coordinate-only work of batch t+1 on a side stream during step t (the headline's protocol); "
                                    "ms_per_step_inline = the same step with it inline"})
        if row["roofline"]:
            row["roofline"]["traffic"], row["roofline"]["traffic_source"] = traffic_of(key, row["roofline"])
        return row
    QUIET, out = True, []
    cpu_before, CPU_BASELINE = CPU_BASELINE, bool(cpu_baselines)
    del CPU_JOBS[:]
    try:
        for key in keys:
            # one failing auxiliary workload (an allocation failure, a kernel error in a less exercised network) must not take the
            # already measured headline line down with it: the row carries the error instead
            try:
                row = one(key)
            except Exception as e:                                  # noqa: BLE001 -- reported, not swallowed
                import traceback
                where = "; ".join(f"{os.path.basename(f.filename)}:{f.lineno} {f.name}" for f in traceback.extract_tb(e.__traceback__)[-4:])
                row = {"key": key, "error": f"{type(e).__name__}: {e}"[:300], "where": where}
            if row is not None:
                out.append(row)
            try:
                torch.cuda.empty_cache()
            except Exception:                                       # noqa: BLE001
                pass
        # the CPU legs after ALL GPU rows (their 32 threads would compete with the launch thread of the host-sensitive rows)
        for r, kind, state in CPU_JOBS:
            try:
                r["cpu_baseline"] = cpu_baseline_of(kind, state, n_warm=1, n_steps=3)
            except Exception as e:                                  # noqa: BLE001
                r["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        for row in out:
            r = row.pop("_r", None)
            if r is not None and "cpu_baseline" in r:
                row["cpu_baseline"] = r["cpu_baseline"]
    finally:
        QUIET, CPU_BASELINE = False, cpu_before
        del CPU_JOBS[:]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--windows", type=int, default=3, help="consecutive timed windows of --steps steps per config; the row reports the median and lists all")
    ap.add_argument("--out", default=None)
    ap.add_argument("--cpu-baseline", action="store_true", help="also time the CPU restatement of configs 2', 3, 4, 5 (tens of seconds each)")
    ap.add_argument("--table", type=int, default=0, help="print the N most expensive (entry point, launch shape) rows per config to stderr")
    ap.add_argument("--only", default=None, help="run only the configs whose name contains this (one config under rocprofv3)")
    ap.add_argument("--dump-launch-order", default=None, help="with --only: write the (entry point, shape, kernel) sequence of one step as JSON (tools/pmc_traffic.py)")
    a = ap.parse_args()
    global ONLY, CPU_BASELINE, TABLE, ORDER_OUT, N_WINDOWS
    ONLY, CPU_BASELINE, TABLE, ORDER_OUT, N_WINDOWS = a.only, a.cpu_baseline, a.table, a.dump_launch_order, max(1, a.windows)
    from pointcloudlib_amd.networks.cls.pointnet import PointNet
    from pointcloudlib_amd.networks.cls.pointnet2 import PointNet2_cls
    from pointcloudlib_amd.networks.cls.dgcnn import DGCNN
    from pointcloudlib_amd.networks.cls.pointconv import PointConvDensityClsSsg
    from pointcloudlib_amd.networks.seg.pointnet2_partseg import PointNet2_partseg
    dev = "cuda"
    def cloud(B, N, seed): return torch.from_numpy(synth.gauss_ball(B, N, seed)).to(dev)
    def nrm(B, N, seed): return torch.from_numpy(synth.unit_normals(B, N, seed)).to(dev)
    def lab(B, seed): return torch.from_numpy(synth.labels(B, 40, seed)).to(dev)
    res = []
    y8, y32 = lab(8, 1), lab(32, 1)
    x = cloud(8, 1024, 20241)
    res.append(run("cfg1 PointNet cls B=8 N=1024", PointNet, (x.transpose(1, 2).contiguous(),), lambda o: soft_cross_entropy_loss(o, y8), a.steps, cpu_kind="cfg1"))
    x = cloud(32, 1024, 20242)
    res.append(run("cfg2 PointNet++ SSG cls B=32 N=1024 (no sampling prefetch)", PointNet2_cls, (x, nrm(32, 1024, 7)), lambda o: soft_cross_entropy_loss(o, y32), a.steps))
    x4 = cloud(32, 4096, 20242)
    res.append(run("cfg2' PointNet++ SSG cls B=32 N=4096", PointNet2_cls, (x4, nrm(32, 4096, 7)), lambda o: soft_cross_entropy_loss(o, y32), a.steps, cpu_kind="cfg2_n4096"))
    res.append(run("cfg3 DGCNN cls B=32 N=1024 k=20", DGCNN, (x.transpose(1, 2).contiguous(),), lambda o: soft_cross_entropy_loss(o, y32), a.steps, cpu_kind="cfg3"))
    xs = cloud(16, 2048, 20244)
    oh = torch.zeros(16, 16, device=dev); oh[torch.arange(16), torch.arange(16) % 16] = 1
    seg = torch.randint(0, 50, (16, 2048), device=dev)
    res.append(run("cfg4-ssg PointNet++ SSG part-seg B=16 N=2048 (the variant train_partseg.py wires)", PointNet2_partseg, (xs, xs, oh),
                   lambda o: seg_cross_entropy_loss(o, seg), a.steps, cpu_kind="cfg4"))
    res.append(run_prefetch("cfg4-ssg PointNet++ SSG part-seg B=16 N=2048, sampling of batch t+1 on a side stream", PointNet2_partseg, (xs, xs, oh),
                            lambda o: seg_cross_entropy_loss(o, seg), a.steps))
    from pointcloudlib_amd.networks.seg.pointnet2_partseg import PointNetMSG
    res.append(run("cfg4 PointNet++ MSG part-seg B=16 N=2048 (BASELINE configs[3]; FP widths corrected, see DESIGN 7)", PointNetMSG, (xs, xs, oh),
                   lambda o: seg_cross_entropy_loss(o, seg), a.steps, cpu_kind="cfg4_msg"))
    res.append(run_prefetch("cfg4 PointNet++ MSG part-seg B=16 N=2048, sampling of batch t+1 on a side stream", PointNetMSG, (xs, xs, oh),
                            lambda o: seg_cross_entropy_loss(o, seg), a.steps))
    res.append(run("cfg5 PointConv cls B=32 N=1024", PointConvDensityClsSsg, (x.transpose(1, 2).contiguous(),), lambda o: soft_cross_entropy_loss(o, y32), a.steps, cpu_kind="cfg5"))
    res.append(run_prefetch("cfg5 PointConv cls B=32 N=1024, densities / FPS / k-NN groups of batch t+1 on a side stream", PointConvDensityClsSsg,
                            (x.transpose(1, 2).contiguous(),), lambda o: soft_cross_entropy_loss(o, y32), a.steps))
    from pointcloudlib_amd.networks.cls.pointcnn import PointCNNcls
    from pointcloudlib_amd.networks.seg.pointcnn_partseg import PointCNN_partseg
    from pointcloudlib_amd.networks.seg.pointnet_partseg import PointNet_partseg
    from pointcloudlib_amd.networks.seg.dgcnn_partseg import DGCNN_partseg
    from pointcloudlib_amd.networks.seg.pointconv_partseg import PointConvDensity_partseg
    ce = seg_cross_entropy_loss
    xst = xs.transpose(1, 2).contiguous()
    res.append(run("PointCNN cls B=32 N=1024", PointCNNcls, (x,), lambda o: soft_cross_entropy_loss(o, y32), a.steps))
    res.append(run("PointNet part-seg B=16 N=2048", PointNet_partseg, (xst, oh), lambda o: ce(o, seg), a.steps))
    res.append(run("DGCNN part-seg B=16 N=2048 k=40", lambda: DGCNN_partseg(50), (xst, oh), lambda o: ce(o, seg), a.steps))
    res.append(run("PointCNN part-seg B=16 N=2048", PointCNN_partseg, (xs,), lambda o: ce(o, seg), a.steps))
    res.append(run("PointConv part-seg B=16 N=2048", PointConvDensity_partseg, (xs, oh), lambda o: ce(o.permute(0, 2, 1), seg), a.steps))
    res = [r for r in res if r is not None]
    for r, kind, state in CPU_JOBS:
        r["cpu_baseline"] = cpu_baseline_of(kind, state)
        print(json.dumps({"config": r["config"], "cpu_baseline": r["cpu_baseline"]}), flush=True)
    if a.out:
        json.dump({"device": torch.cuda.get_device_name(0), "note": "1 GPU, fp32, synthetic gauss_ball clouds, fwd+bwd+SGD; roofline = the dominant own kernel of each config (bench.py's rule)", "results": res},
                  open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
