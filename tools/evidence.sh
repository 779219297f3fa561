#!/bin/bash
# One GPU-box pass that regenerates the measured evidence of a round (run through gpurun; copies go to profiles/ by hand):
#   bash tools/evidence.sh r04
# kernel trace + stats, FETCH_SIZE / WRITE_SIZE passes (separate runs, counters only), SQ wave-state pass, the bench line.
R=${1:-r04}; O=gpurun_out/$R; mkdir -p $O; cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 3 --no-settle --no-cpu-baseline --no-other-configs --windows 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o b -- $B > $O/kt.log 2>&1
S="python bench.py --steps 3 --warmup 1 --no-settle --no-cpu-baseline --no-other-configs --roofline-kernel none --windows 1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o b -- $S --dump-launch-order $O/order.json > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o b -- $S > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d $O/sq -o b -- $S > $O/sq.log 2>&1
F=$(find $O/fetch -name '*counter_collection.csv' | head -1); W=$(find $O/write -name '*counter_collection.csv' | head -1)
python tools/pmc_traffic.py $F $W $O/order.json $O/${R}_pmc_hbm_traffic.csv $O/${R}_traffic.json > $O/traffic.txt 2>&1
python tools/pmc_sq.py $(find $O/sq -name '*counter_collection.csv' | head -1) $O/${R}_pmc_sq_wave_states.csv
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/${R}_bench_kernel_stats.csv
T=$(find $O/kt -name '*kernel_trace.csv' | head -1)
python tools/timeline.py $T > $O/${R}_step_timeline.csv 2>&1
python tools/kernel_by_shape.py $T 'linear_bwd_fused_kernel<true, true, 4, 2, false>' 1 0=fb256x128 > $O/${R}_dominant_kernel_by_shape.csv 2>&1
python tools/kernel_by_shape.py $T 'linear_bwd_fused_kernel<true, true, 2, 1, true>' 1 0=fb128x64 | tail -1 >> $O/${R}_dominant_kernel_by_shape.csv
python tools/kernel_by_shape.py $T 'linear_bwd_fused_kernel<false, true, 2, 2, false>' 1 0=fb128x128 | tail -1 >> $O/${R}_dominant_kernel_by_shape.csv
python tools/kernel_by_shape.py $T 'linear_bwd_fused_kernel<false, true, 1, 1, false>' 1 0=fb64x64 | tail -1 >> $O/${R}_dominant_kernel_by_shape.csv
rm -rf $O/kt/*/*.db $O/fetch $O/write $O/sq/*/*.db 2>/dev/null
# (the bench line is taken at the END of this script, after every traffic profile of THESE sources has been copied to profiles/ on this box)
# every C-ABI entry point event-timed (GEMM kernels: their own begin/end timestamps), algorithmic GB/s and TF per launch shape
python bench.py --steps 20 --warmup 3 --profile-all --no-cpu-baseline --no-other-configs --roofline-kernel none 2>&1 >/dev/null | grep -E "n/step|entry" > $O/${R}_entry_point_roofline.txt
python -m pytest tests/test_parity_pointnet_gpu.py tests/test_parity_pointnet2_gpu.py tests/test_parity_dgcnn_gpu.py tests/test_parity_partseg_gpu.py tests/test_parity_pointconv_gpu.py tests/test_pointcnn_gpu.py tests/test_parity_partseg_zoo_gpu.py -k "not xconv and not stage and not partseg_network and not cls_network" -m gpu -s -q 2>&1 | grep -v Warning > $O/${R}_parity_reports.txt
# any parity row that moved by more than 10 % against the previous round's committed report
PREV=$(ls profiles/r0*_parity_reports.txt 2>/dev/null | grep -v "${R}_" | sort | tail -1)
[ -n "$PREV" ] && python tools/parity_diff.py $PREV $O/${R}_parity_reports.txt > $O/${R}_parity_rows_moved.txt 2>&1
# per-phase cycles of the fused backward's tile loop (lab build EXP=7: built beforehand with `make -C pointcloudlib_amd/csrc EXP=7`)
if [ -f pointcloudlib_amd/libpcl_hip_exp7.so ]; then
  PCL_HIP_SO=$PWD/pointcloudlib_amd/libpcl_hip_exp7.so python bench.py --steps 3 --warmup 2 --no-settle --no-cpu-baseline --no-other-configs --roofline-kernel none 2>&1 | grep "^fb<" | sort | uniq -c | sort -rn | awk '{$1=""; print}' | sort | awk 'NR%5==1' > $O/${R}_fused_backward_phase_cycles.txt
fi
# the other BASELINE configs: one line each with the roofline of its dominant kernel; per config kernel stats + PMC HBM traffic
python tools/bench_models.py --steps 20 --cpu-baseline --out $O/${R}_other_configs.json > $O/other.log 2>&1
bash tools/traffic_cfg.sh $R cfg2_n4096 "cfg2'" > $O/tc_cfg2.log 2>&1
bash tools/traffic_cfg.sh $R cfg3 "cfg3" > $O/tc_cfg3.log 2>&1
bash tools/traffic_cfg.sh $R cfg4 "cfg4 PointNet++ MSG part-seg B=16 N=2048 (BASELINE" > $O/tc_cfg4.log 2>&1
bash tools/traffic_cfg.sh $R cfg5 "cfg5 PointConv cls B=32 N=1024$" > $O/tc_cfg5.log 2>&1
cp $O/${R}_traffic.json $O/${R}_traffic_cfg*.json profiles/ 2>/dev/null
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${R}_bench_line.json 2> $O/bench.err      # the driver's exact command
tail -3 $O/kt.log; cat $O/traffic.txt | head -40; cat $O/${R}_dominant_kernel_by_shape.csv; tail -1 $O/${R}_bench_line.json
