#!/bin/bash
# per-dispatch durations of the kernels whose name contains $2, one bench_models config ($1 = --only substring) under rocprofv3 --kernel-trace
O=gpurun_out/kc; mkdir -p $O; cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o b -- python tools/bench_models.py --steps 6 --only "$1" > $O/kt.log 2>&1
T=$(find $O/kt -name '*kernel_trace.csv' | head -1)
python - "$T" "$2" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
d = collections.defaultdict(list)
for r in rows:
    key = (r["Kernel_Name"][:70], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r.get("Grid_Size_Y", ""))
    d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v.sort()
    print(f"{k[0]} grid {k[1]}x{k[2]}: n={len(v)} median {v[len(v)//2]:.1f} us min {v[0]:.1f} max {v[-1]:.1f}")
PY
rm -rf $O/kt
