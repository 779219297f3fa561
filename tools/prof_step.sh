#!/bin/bash
# kernel trace + stats of the headline bench and the timeline of one step (run through gpurun):  bash tools/prof_step.sh <tag>
R=${1:-x}; O=gpurun_out/$R; mkdir -p $O; cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 3 --no-settle --no-cpu-baseline --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o b -- $B > $O/kt.log 2>&1
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/${R}_bench_kernel_stats.csv
T=$(find $O/kt -name '*kernel_trace.csv' | head -1)
python tools/timeline.py $T > $O/${R}_step_timeline.csv 2>&1
rm -rf $O/kt
tail -2 $O/kt.log
python bench.py --no-cpu-baseline --no-other-configs > $O/${R}_bench_line.json 2> $O/bench.err; tail -1 $O/${R}_bench_line.json
