#!/bin/bash
# The ISA lens (tools/isa_lens.py) over the headline step's own kernels: runs on the build host (hipcc -S, no GPU).
#   bash tools/isa_report.sh > profiles/r06_isa_lens.txt
cd "$(dirname "$0")/.."
C=pointcloudlib_amd/csrc
echo "# tools/isa_lens.py over the kernels of the headline step (hipcc -S --offload-arch=gfx950, static counts per kernel body; loops are mostly unrolled,"
echo "# so 'per kernel' reads as 'per row tile' for the persistent GEMM kernels).  What the columns mean: tools/isa_lens.py's docstring."
for spec in "mlp.hip linear_fwd_res_kernel" "mlp.hip linear_bwd_fused_kernel" "mlp.hip bn_act_max_rows_kernel" "compact.hip group_linear_v4_kernel group_linear_bwd_v4_kernel group_linear_bwd_gather_kernel" "fps.hip fps_kernelILi256ELi4 fps_kernelILi64ELi8" "ball_query.hip ball_query_kernel"; do
  set -- $spec; f=$1; shift
  echo; echo "## $f: $*"
  python3 tools/isa_lens.py $C/$f "$@" | c++filt | cut -c1-400
done
