#!/bin/bash
# kernel stats of one bench_models config under rocprofv3:  bash tools/prof_cfg.sh <tag> "<--only substring>"
R=${1:-x}; O=gpurun_out/$R; mkdir -p $O; cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o b -- python tools/bench_models.py --steps 10 --only "$2" > $O/kt.log 2>&1
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/${R}_kernel_stats.csv
rm -rf $O/kt
grep config $O/kt.log | cut -c1-200
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/${R}_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}%  calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"])/1e3:8.1f} us  {r["Name"][:110]}')
PY
