#!/bin/bash
# DESIGN 9.8's numbers in one file: per-phase cycles of the bf16-plane forward (make EXP=8), the step and the GEMM entry points under
# both matrix forms, the staged GEMM in both forms, HBM write bandwidth.   bash tools/dbg/split_evidence.sh > profiles/r04_split_bf16_experiment.txt
echo "== per-phase cycles per tile of linear_fwd_split_kernel (EXP=8 build; waves 0 = early, 7 = late), PCL_MATRIX_FORM=1"
PCL_MATRIX_FORM=1 PCL_HIP_SO=$PWD/pointcloudlib_amd/libpcl_hip_exp8.so python bench.py --no-cpu-baseline --no-other-configs --roofline-kernel none --steps 3 --warmup 2 2>&1 | grep "^fs<" | sort | awk '{k=$1" "$3; n[k]++; if (n[k]<=2) print}'
for f in 0 1; do
  echo "== PCL_MATRIX_FORM=$f: forward entry points (us per launch, event-timed) and the step"
  PCL_MATRIX_FORM=$f python bench.py --no-cpu-baseline --no-other-configs --profile-all --steps 20 2>&1 | grep "fwd[0-9]*x[0-9]* \|ms_per_step" | sed 's/.*"ms_per_step": \([0-9.]*\).*/ms_per_step \1/' | cut -c1-120
done
echo "== three interleaved pairs of the step (ms): default | PCL_MATRIX_FORM=1"
bash tools/ab.sh "PCL_MATRIX_FORM=1" 2>&1 | tail -6
echo "== staged GEMM forward, fp32 MFMA (form 0) vs bf16 planes (form 0x1002)"
python tools/dbg/split_gemm_time.py 2>&1 | grep "^rows"
echo "== HBM streaming (tools/ubench/write_bw.hip)"
[ -x tools/ubench/write_bw.bin ] && ./tools/ubench/write_bw.bin | grep "grid  1024\|grid  4096"
