#!/bin/bash
# The driver's exact headline command, N times on this box: the distribution of the line.
#   bash tools/driver_dist.sh r06a [N_full] [N_short]     (through gpurun; one box per call -- repeat on several boxes)
# Full runs = `python3 bench.py --gpus 1 --steps 20 --warmup 5` exactly; short runs add --no-cpu-baseline --no-other-configs (the
# timed region is over before those legs start, so its numbers are the same experiment at a fifth of the box time).
R=${1:-r06a}; NF=${2:-2}; NS=${3:-8}; O=gpurun_out/$R; mkdir -p $O
{ echo "numa nodes of the GPUs: $(cat /sys/class/drm/renderD*/device/numa_node 2>/dev/null | tr '\n' ' ')"; lscpu | grep -E "Model name|NUMA node|Socket" ; } > $O/box.txt
i=0
for k in $(seq 1 $NF); do i=$((i+1)); python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/line_$i.json 2> $O/err_$i.txt; done
for k in $(seq 1 $NS); do i=$((i+1)); python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/line_$i.json 2> $O/err_$i.txt; done
# the other NUMA node and no pinning at all (two short runs each)
other=$(python3 - <<'PY'
import sys; sys.path.insert(0, ".")
from pointcloudlib_amd.affinity import gpu_numa_nodes
g = gpu_numa_nodes(); print(1 - g[0] if g and g[0] in (0, 1) else 0)
PY
)
for k in 1 2; do i=$((i+1)); PCL_PIN_NODE=$other python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/line_other_node_$k.json 2> $O/err_o$k.txt; done
for k in 1 2; do i=$((i+1)); PCL_PIN_NODE=-1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/line_unpinned_$k.json 2> $O/err_u$k.txt; done
python3 - $O <<'PY'
import glob, json, sys, os
O = sys.argv[1]
rows = []
for f in sorted(glob.glob(O + "/line_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        rows.append(f"{os.path.basename(f)}: no line ({e})"); continue
    rows.append(f"{os.path.basename(f):24s} ms_per_step {d['ms_per_step']:.4f}  windows {d.get('windows_ms_per_step')}  armed {d.get('windows_timer_armed')}  "
                f"host_in_window {d.get('windows_host_enqueue_ms_per_step')}  host_enqueue {d.get('host_enqueue_ms')}  affinity {d['config'].get('cpu_affinity')}  "
                f"kernel {d['roofline']['avg_launch_ms'] if d.get('roofline') else None}")
open(O + "/summary.txt", "w").write(open(O + "/box.txt").read() + "\n".join(rows) + "\n")
print(open(O + "/summary.txt").read())
PY
