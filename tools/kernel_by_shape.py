"""rocprofv3 --kernel-trace gives one row per dispatch; --stats averages a kernel NAME, which for the GEMM template mixes
the launch shapes that share an instantiation.  This splits a kernel's dispatches by their slot within a step (launch
order is the same every step) and prints the average per slot.
    python tools/kernel_by_shape.py <b_kernel_trace.csv> '<substring of the kernel name>' <launches per step> [slot=shape ...]"""
import csv, sys
path, name, per = sys.argv[1], sys.argv[2], int(sys.argv[3])
labels = dict(a.split("=") for a in sys.argv[4:])
rows = [r for r in csv.DictReader(open(path)) if name in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
assert rows and len(rows) % per == 0, (len(rows), per)
print("kernel,slot_in_step,shape,dispatches,avg_us,min_us,max_us")
for s in range(per):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[s::per]]
    print(f'"{rows[0]["Kernel_Name"]}",{s},{labels.get(str(s), "")},{len(d)},{sum(d) / len(d):.1f},{min(d):.1f},{max(d):.1f}')
