#!/bin/bash
# A/B of host-side lab switches (environment variables read by pointcloudlib_amd/_lib.py) on ONE box, interleaved:
#   bash tools/ab_env.sh OUT "grep pattern of the entry-point table" name1=VAR=val name2=VAR=val ...
O=$1; PAT=$2; shift 2; mkdir -p $O
for rep in 1 2 3; do
  for v in "$@"; do
    n=${v%%=*}; kv=${v#*=}
    env $kv python bench.py --no-cpu-baseline --no-other-configs --steps 20 --roofline-kernel none 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('rep $rep [$n]', d['ms_per_step'], d['windows_ms_per_step'])"
  done
done
for v in "$@"; do
  n=${v%%=*}; kv=${v#*=}
  echo "--- $n"
  env $kv python bench.py --steps 10 --warmup 3 --windows 1 --profile-all --no-cpu-baseline --no-other-configs --roofline-kernel none 2>&1 >/dev/null | grep -E "n/step" | grep -E "$PAT"
done
