#!/bin/bash
# A/B of library builds on ONE box: bash tools/ab_so.sh OUT name1=path1.so name2=path2.so ...   (bench.py headline, interleaved, 2 reps; then
# the per-entry-point table of each build)
O=$1; shift; mkdir -p $O
for rep in 1 2; do
  for v in "$@"; do
    n=${v%%=*}; so=${v#*=}
    PCL_HIP_SO=$PWD/$so python bench.py --no-cpu-baseline --no-other-configs --steps 20 2>/dev/null > $O/ab_${n}_$rep.json
    python - $O/ab_${n}_$rep.json $n $rep <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(f"rep {sys.argv[3]} [{sys.argv[2]:8s}] median {d['ms_per_step']:.4f} windows {d['windows_ms_per_step']} dominant {d['roofline']['shape']} {d['roofline']['avg_launch_ms']}")
PY
  done
done
for v in "$@"; do
  n=${v%%=*}; so=${v#*=}
  PCL_HIP_SO=$PWD/$so python bench.py --steps 10 --warmup 3 --windows 1 --profile-all --no-cpu-baseline --no-other-configs --roofline-kernel none 2>&1 >/dev/null | grep -E "n/step" | grep -E "fb|fused" > $O/table_$n.txt
  echo "--- $n"; cat $O/table_$n.txt
done
