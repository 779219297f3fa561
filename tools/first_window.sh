#!/bin/bash
# Why is the first timed window of bench.py 4-5 % slower than the four behind it (14 of 14 runs, gpurun_out/r06a)?  Lab switches:
#   PCL_BENCH_NO_GC=1     no gc.collect() / gc.disable() in front of the windows
#   PCL_BENCH_IDLE_MS=n   n ms of host sleep with the GPU idle in front of window 2 (does an idle GPU make the NEXT window slow?)
R=${1:-r06b}; O=gpurun_out/$R; mkdir -p $O
B="python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs"
for k in 1 2; do $B > $O/base_$k.json 2>/dev/null; done
for k in 1 2; do PCL_BENCH_NO_GC=1 $B > $O/nogc_$k.json 2>/dev/null; done
for ms in 20 100 400; do PCL_BENCH_IDLE_MS=$ms $B > $O/idle${ms}.json 2>/dev/null; done
PCL_BENCH_NO_GC=1 PCL_BENCH_IDLE_MS=100 $B > $O/nogc_idle100.json 2>/dev/null
$B --roofline-kernel none > $O/notimer.json 2>/dev/null
$B --windows 9 > $O/nine.json 2>/dev/null
python3 - $O <<'PY'
import glob, json, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f"{os.path.basename(f):22s} median {d['ms_per_step']:.4f}  windows {d['windows_ms_per_step']}  host {d['windows_host_enqueue_ms_per_step']}")
    except Exception as e:
        print(os.path.basename(f), "no line", e)
PY
