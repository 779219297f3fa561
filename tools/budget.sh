#!/bin/bash
# Round 5: whole-kernel cycle budget of the fused backward (lab build EXP=7: time stamps in a device buffer, no printf) beside the
# clock micro-benchmark, GRBM_GUI_ACTIVE and the kernel trace of the product build on the SAME box:   bash tools/budget.sh r05
R=${1:-r05}; O=gpurun_out/$R; mkdir -p $O; cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/ubench/clock.bin > $O/${R}_clock_ubench.txt 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/clk -o c -- tools/ubench/clock.bin > $O/clk.log 2>&1
python - $O >> $O/${R}_clock_ubench.txt <<'PY'
import csv, sys, glob
O = sys.argv[1]
cc = glob.glob(O + "/clk/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(O + "/clk/**/*kernel_trace.csv", recursive=True)[0]
dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(kt))}
print("# rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -- tools/ubench/clock.bin : GUI_ACTIVE / 8 XCDs per dispatch against the dispatch's duration")
n = 0
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    n += 1
    if "empty" in r["Kernel_Name"] and n > 20: continue
    d = dur.get(r["Dispatch_Id"], 0)
    g = float(r["Counter_Value"]) / 8
    print(f'{r["Kernel_Name"][:40]:40s} grid {r.get("Grid_Size", r.get("Grid_Size_X"))}: GUI_ACTIVE/8 {g:12.0f}  duration {d/1e3:9.1f} us  ratio {g/max(d,1):.3f} GHz')
PY
python tools/fb_budget.py run $O/fbk.bin > $O/exp7.log 2>&1
python tools/fb_budget.py show $O/fbk.bin > $O/${R}_fb_budget_raw.txt 2>&1
S="python bench.py --steps 3 --warmup 2 --no-settle --no-cpu-baseline --no-other-configs --roofline-kernel none --windows 1"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \
    --output-format csv -d $O/sq -o b -- $S > $O/sq.log 2>&1
python tools/pmc_sq.py $(find $O/sq -name '*counter_collection.csv' | head -1) $O/${R}_pmc_sq_wave_states.csv > $O/sq_tool.log 2>&1
# the same dispatches' durations (profiled pass): GUI_ACTIVE / 8 / duration per fused-backward launch
python - $O >> $O/${R}_fb_budget_raw.txt <<'PY'
import csv, sys, glob, statistics as st
O = sys.argv[1]
cc = glob.glob(O + "/sq/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(O + "/sq/**/*kernel_trace.csv", recursive=True)[0]
dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(kt))}
acc = {}
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or "linear_bwd_fused" not in r["Kernel_Name"]: continue
    k = r["Kernel_Name"].split("(")[0]
    acc.setdefault(k, []).append((float(r["Counter_Value"]) / 8, dur[r["Dispatch_Id"]] / 1e3))
print("=" * 118)
print("# product build under rocprofv3 --pmc (same box): GRBM_GUI_ACTIVE / 8 XCDs per launch against that launch's own duration in the trace")
for k, v in acc.items():
    print(f"{k}: GUI_ACTIVE/8 median {st.median(x[0] for x in v):.0f}, duration median {st.median(x[1] for x in v):.1f} us, ratio {st.median(x[0]/x[1]/1e3 for x in v):.3f} 'GHz'")
PY
B="python bench.py --steps 20 --warmup 3 --no-settle --no-cpu-baseline --no-other-configs --windows 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o b -- $B > $O/kt.log 2>&1
T=$(find $O/kt -name '*kernel_trace.csv' | head -1)
python tools/kernel_by_shape.py $T 'linear_bwd_fused_kernel<true, true, 4, 2, false>' 1 0=fb256x128 > $O/${R}_dominant_kernel_by_shape.csv 2>&1
python tools/kernel_by_shape.py $T 'linear_bwd_fused_kernel<true, true, 2, 1, true>' 1 0=fb128x64 | tail -1 >> $O/${R}_dominant_kernel_by_shape.csv
python tools/kernel_by_shape.py $T 'linear_bwd_fused_kernel<false, true, 2, 2, false>' 1 0=fb128x128 | tail -1 >> $O/${R}_dominant_kernel_by_shape.csv
python tools/kernel_by_shape.py $T 'linear_bwd_fused_kernel<false, true, 1, 1, false>' 1 0=fb64x64 | tail -1 >> $O/${R}_dominant_kernel_by_shape.csv
rm -rf $O/kt/*/*.db $O/sq/*/*.db $O/clk/*/*.db 2>/dev/null
cat $O/${R}_clock_ubench.txt; cat $O/${R}_fb_budget_raw.txt; cat $O/${R}_dominant_kernel_by_shape.csv; tail -2 $O/exp7.log
