#!/bin/bash
# A/B timing of variant builds (csrc/Makefile EXP=n) on the GPU box: per-entry-point table of one bench step per variant.
#   gpurun -- 'bash tools/lab.sh "" 1 2 3'      ("" = the product library)
out=gpurun_out/lab; mkdir -p $out
for e in "$@"; do
  so=pointcloudlib_amd/libpcl_hip${e:+_exp$e}.so
  PCL_HIP_SO=$PWD/$so python bench.py --steps 20 --warmup 3 --profile-all --no-cpu-baseline --roofline-kernel none > $out/exp_${e:-base}.log 2>&1
  echo "=== EXP ${e:-base}: $(tail -1 $out/exp_${e:-base}.log | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])' 2>/dev/null)"
  grep -E "pcl_linear" $out/exp_${e:-base}.log | awk '{printf "%s %s %s | ", $2, $5, $10}' ; echo
done
