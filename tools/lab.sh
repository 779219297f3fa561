#!/bin/bash
# A/B timing on the GPU box: per-entry-point table of one bench step per variant.
#   bash tools/lab.sh VAR=VALUE[,VAR=VALUE...] ...     each argument = one run with those environment variables ("-" = none)
#   (a variant library built with `make -C pointcloudlib_amd/csrc EXP=n` is selected with PCL_HIP_SO=$PWD/pointcloudlib_amd/libpcl_hip_expN.so)
out=gpurun_out/lab; mkdir -p $out
for e in "$@"; do
  tag=$(echo "$e" | tr -c 'A-Za-z0-9=,._\n-' '_')
  envs=$(echo "$e" | tr ',' ' '); [ "$e" = "-" ] && envs=""
  env $envs python bench.py --steps 20 --warmup 3 --profile-all --no-cpu-baseline --no-other-configs --roofline-kernel none > $out/$tag.log 2>&1
  echo "=== $e: ms/step $(tail -1 $out/$tag.log | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])' 2>/dev/null)"
  grep -E "pcl_linear" $out/$tag.log | awk '{printf "%s %.0fus %sTF | ", $2, $6*1000, $10}' ; echo
done
