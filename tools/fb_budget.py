"""Whole-kernel cycle budget of the fused backward (round 5) from the lab build's time stamps (make EXP=7, csrc/mlp.hip: g_fbk).

    python tools/fb_budget.py run OUT.bin      # on the GPU box: bench steps on libpcl_hip_exp7.so, then the stamps of the last 8 launches
    python tools/fb_budget.py show OUT.bin     # the table

Every workgroup's wave 0 and wave 7 record the shader-clock counter (s_memtime; tools/ubench/clock.hip shows it IS the shader clock) at
entry / after the prologue / after the tile loop / after the partial-tile stores are issued / after they have drained, the constant
100 MHz counter (s_memrealtime) at entry and exit, and the per-phase sums of the tile loop.
"""
import os
import statistics as st
import sys

import numpy as np

SHAPES = ["sparse 256x128 (SA2 layer 3, the dominant kernel)", "sparse 128x64 (SA1 layer 3)", "dense 128x128 (SA2 layer 2)", "dense 64x64 (SA1 layer 2)"]
RING, WORDS = 8, 16


def run(out):
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.environ["PCL_HIP_SO"] = os.path.join(ROOT, "pointcloudlib_amd", "libpcl_hip_exp7.so")
    sys.path.insert(0, ROOT)
    sys.argv = ["bench.py", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-other-configs", "--roofline-kernel", "none"]
    import ctypes
    import bench
    bench.main()
    from pointcloudlib_amd import _lib
    L = ctypes.CDLL(_lib.so_path())
    L.pcl_lab_fbk_read.restype = ctypes.c_int
    L.pcl_lab_fbk_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    n = 4 * RING * 256 * 2 * WORDS * 8
    buf = ctypes.create_string_buffer(n)
    rc = L.pcl_lab_fbk_read(buf, n)
    assert rc == 0, rc
    open(out, "wb").write(buf.raw)


def show(path):
    a = np.frombuffer(open(path, "rb").read(), dtype=np.int64).reshape(4, RING, 256, 2, WORDS)
    med = lambda v: float(st.median(v))
    for sh in range(4):
        launches = [a[sh, r] for r in range(RING) if a[sh, r, 0, 0, 1] > 0]
        if not launches:
            continue
        print("=" * 118)
        print("fused backward, %s -- %d launches" % (SHAPES[sh], len(launches)))
        agg = {}
        add = lambda k, v: agg.setdefault(k, []).append(v)
        for l in launches:
            w0 = l[:, 0, :]
            live = w0[:, 14] > 0
            w0 = w0[live]
            rt0, rt1, tk = w0[:, 0], w0[:, 1], w0[:, 2:7]
            tot = tk[:, 4] - tk[:, 0]
            clk = tot / ((rt1 - rt0) * 10.0)
            add("wg", len(w0)); add("clk", med(clk)); add("clk_lo", clk.min()); add("clk_hi", clk.max())
            # per XCD (workgroup b runs on XCD b % 8): is the clock per die?
            add("clk_xcd", [med(clk[np.arange(len(clk)) % 8 == x]) for x in range(8)])
            add("wall_us", (rt1.max() - rt0.min()) / 100.0); add("skew_us", (rt0.max() - rt0.min()) / 100.0)
            ex = np.sort(rt1)
            add("exit_spread", (ex[-1] - ex[0]) / 100.0); add("exit_med_last", (ex[-1] - ex[len(ex) // 2]) / 100.0)
            tiles = w0[:, 14]
            add("tmin", tiles.min()); add("tmax", tiles.max()); add("nmax", int((tiles == tiles.max()).sum()))
            for i, k in enumerate(("pro", "loop", "wo", "drain")):
                d = tk[:, i + 1] - tk[:, i]
                add(k + "_med", med(d)); add(k + "_max", d.max())
            add("tot_med", med(tot)); add("tot_max", tot.max())
            big = tiles == tiles.max()
            add("loop_tile", med((tk[big, 2] - tk[big, 1]) / tiles[big]))
            last = np.argmax(rt1)
            add("last_tot", tot[last]); add("last_tiles", tiles[last])
            for wv in (0, 1):
                ph = l[:, wv, 7:14][live][big] / tiles[big][:, None]
                add("ph%d" % wv, np.median(ph, axis=0))
        m = lambda k: med(agg[k])
        print("  shader clock = s_memtime ticks / (s_memrealtime ticks x 10 ns)  : %.3f GHz median over workgroups (min %.3f, max %.3f); per XCD: %s"
              % (m("clk"), m("clk_lo"), m("clk_hi"), " ".join("%.2f" % x for x in np.median(np.array(agg["clk_xcd"]), axis=0))))
        print("  launch wall, first workgroup's entry -> last one's exit          : %.1f us = %.0f k cycles at that clock" % (m("wall_us"), m("wall_us") * m("clk")))
        print("  entry skew %.2f us; exit spread first -> last %.1f us, median -> last %.1f us" % (m("skew_us"), m("exit_spread"), m("exit_med_last")))
        print("  tiles per workgroup %d .. %d (%d of %d workgroups have the maximum)" % (m("tmin"), m("tmax"), m("nmax"), m("wg")))
        print("  per workgroup, wave 0, shader cycles                                  median      max over workgroups")
        for k, lab in (("pro", "prologue: constants, first row records, first tile's requests"), ("loop", "tile loop"),
                       ("wo", "partial dW tile(s) + BatchNorm sums: stores issued"), ("drain", "... and drained (s_waitcnt vmcnt(0))"), ("tot", "entry -> exit")):
            print("    %-64s %9.0f   %9.0f" % (lab, m(k + "_med"), m(k + "_max")))
        print("  tile loop per tile (workgroups with the maximum count)            : %.0f cycles;  x %d tiles = %.0f k" % (m("loop_tile"), m("tmax"), m("loop_tile") * m("tmax") / 1e3))
        print("  the LAST workgroup to exit: %d tiles, %.0f k cycles entry -> exit" % (m("last_tiles"), m("last_tot") / 1e3))
        names = ["deposit", "barrier A", "dX", "epilogue", "dW", "barrier B", "top"]
        for wv, nm in ((0, "wave 0 (dX first)"), (1, "wave 7 (dW first)")):
            v = np.median(np.array(agg["ph%d" % wv]), axis=0)
            print("  phases per tile, %s: " % nm + " | ".join("%s %.0f" % (n, x) for n, x in zip(names, v)) + "  = %.0f" % v.sum())


if __name__ == "__main__":
    (run if sys.argv[1] == "run" else show)(sys.argv[2])
