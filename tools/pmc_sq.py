"""Wave-state / MFMA-pipe summary per kernel and grid from one rocprofv3 counter pass:
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \
        SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d out -o b -- python bench.py --steps 3 --warmup 1 --no-settle \
        --no-cpu-baseline --roofline-kernel none
    python tools/pmc_sq.py out/b_counter_collection.csv profiles/rNN_pmc_sq_wave_states.csv"""
import csv, sys
from collections import defaultdict
src, dst = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for r in csv.DictReader(open(src)):
    name = r["Kernel_Name"].split("(")[0]
    key = (name, int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]))
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[key].add(r["Dispatch_Id"])
with open(dst, "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python bench.py --steps 3 --warmup 1 --no-settle\n")
    f.write("# per-launch means.  SQ_WAVE_CYCLES/WAIT_*/ACTIVE_* are quad-cycles (fractions of wave time are unit-free); SQ_VALU_MFMA_BUSY_CYCLES = 64 x (number of v_mfma_f32_32x32x2_f32), summed over the 1024 SIMDs;\n")
    f.write("# GRBM_GUI_ACTIVE is summed over the 8 XCDs: mfma_pipe_util = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024)\n")
    f.write("kernel,grid,launches,wait_any_frac,wait_inst_frac,active_frac,mfma_busy_cycles,gui_active,mfma_pipe_util\n")
    for key in sorted(acc):
        c, k = acc[key], len(n[key])
        wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / k
        mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / k
        util = mf / (gui / 8 * 1024) if gui else 0.0
        f.write(f'"{key[0]}",{key[1]},{k},{c.get("SQ_WAIT_ANY", 0) / wc:.3f},{c.get("SQ_WAIT_INST_ANY", 0) / wc:.3f},'
                f'{c.get("SQ_ACTIVE_INST_ANY", 0) / wc:.3f},{mf:.0f},{gui:.0f},{util:.3f}\n')
