"""One steady-state step out of a rocprofv3 --kernel-trace CSV as a timeline: per dispatch the queue, start offset, duration
and the idle gap since the previous dispatch of the same queue ended; then totals (busy time per queue, gaps, short kernels).
    python tools/timeline.py <kernel_trace.csv> [marker kernel substring = soft_ce_kernel] [which step = -3]
The step runs from one marker kernel to the next."""
import csv, sys
path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "soft_ce_kernel"
which = int(sys.argv[3]) if len(sys.argv) > 3 else -3
rows = list(csv.DictReader(open(path)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = marks[which], marks[which + 1]
step = rows[a:b]
t0 = step[0]["s"]
last_end = {}
qname = {}
busy, gaps, short_t, short_n = {}, {}, 0.0, 0
print(f"# step of {(rows[b]['s'] - t0) / 1e3:.1f} us, {len(step)} dispatches")
print("queue,start_us,dur_us,gap_us,kernel")
for r in step:
    q = qname.setdefault(r["Queue_Id"], len(qname))
    dur = (r["e"] - r["s"]) / 1e3
    gap = (r["s"] - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = r["e"]
    busy[q] = busy.get(q, 0.0) + dur
    gaps[q] = gaps.get(q, 0.0) + max(gap, 0.0)
    if dur < 13.0:
        short_t += dur; short_n += 1
    name = r["Kernel_Name"].replace("void pcl::", "").replace("pcl::", "")
    print(f"{q},{(r['s'] - t0) / 1e3:.1f},{dur:.1f},{gap:.1f},\"{name[:90]}\"")
for q in busy:
    print(f"# queue {q}: busy {busy[q]:.1f} us, idle gaps between its dispatches {gaps[q]:.1f} us")
print(f"# dispatches shorter than 13 us: {short_n}, {short_t:.1f} us")
