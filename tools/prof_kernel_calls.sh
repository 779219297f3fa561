#!/bin/bash
# per-dispatch durations of the kernels matching a substring, one bench_models config:  bash tools/prof_kernel_calls.sh <tag> "<--only>" "<kernel substring>"
R=${1:-x}; O=gpurun_out/$R; mkdir -p $O; cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o b -- python tools/bench_models.py --steps 6 --only "$2" > $O/kt.log 2>&1
T=$(find $O/kt -name '*kernel_trace.csv' | head -1)
python - "$T" "$3" <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
last=rows[-int(len(rows)/ (6+3+2)) :] if rows else []
for r in last:
    print(f'{(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} us  grid {r.get("Grid_Size_X","?")}x{r.get("Grid_Size_Y","?")}  {r["Kernel_Name"][:60]}')
PY
rm -rf $O/kt
