#!/bin/bash
# A/B timing of lab switches on ONE box (boxes differ by +-4 %): bash tools/ab.sh "ENV_A=1" "ENV_B=1" ... ; each variant 3 times, interleaved
for rep in 1 2 3; do
  for v in "" "$@"; do
    r=$(env $v python bench.py --no-cpu-baseline --no-other-configs --roofline-kernel none --steps 60 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "rep $rep [${v:-default}] $r ms"
  done
done
