#!/bin/bash
# Which HIP API calls / memory copies sit between the kernels of one step (run through gpurun): kernel + HIP runtime + memory-copy
# traces of a short bench run (no counters), and the API calls made while the main stream idles at the loss -> backward turn.
R=${1:-x}; O=gpurun_out/$R; mkdir -p $O; cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --hip-runtime-trace --memory-copy-trace --output-format csv -d $O/tr -o b -- python bench.py --steps 6 --warmup 3 --no-settle --no-cpu-baseline --roofline-kernel none > $O/tr.log 2>&1
find $O/tr -name "*.csv" | head
python - <<PY
import csv, glob
k = sorted(csv.DictReader(open(glob.glob("$O/tr/**/*kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
a = list(csv.DictReader(open(glob.glob("$O/tr/**/*hip_api_trace.csv", recursive=True)[0])))
marks = [i for i, r in enumerate(k) if "soft_ce_kernel" in r["Kernel_Name"]]
i0 = marks[-2]
t0 = int(k[i0]["Start_Timestamp"])
# correlate: kernels of the step with their launching API call (Correlation_Id)
api = {r["Correlation_Id"]: r for r in a}
print("kernel start_us dur_us | launch api start_us (host) | thread")
for r in k[i0:i0 + 16]:
    ap = api.get(r["Correlation_Id"])
    print(f'{r["Kernel_Name"][:50]:50s} {(int(r["Start_Timestamp"]) - t0) / 1e3:8.1f} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:7.1f} | '
          + (f'{(int(ap["Start_Timestamp"]) - t0) / 1e3:9.1f} {ap["Function"]} tid {ap["Thread_Id"]}' if ap else "?"))
# all API calls (any thread) in the host-time window of those launches
w0 = min(int(api[r["Correlation_Id"]]["Start_Timestamp"]) for r in k[i0:i0 + 16] if r["Correlation_Id"] in api)
w1 = max(int(api[r["Correlation_Id"]]["End_Timestamp"]) for r in k[i0:i0 + 16] if r["Correlation_Id"] in api)
print("--- API calls in that host window")
for r in sorted(a, key=lambda r: int(r["Start_Timestamp"])):
    s = int(r["Start_Timestamp"])
    if w0 <= s <= w1:
        print(f'{(s - t0) / 1e3:9.1f} +{(int(r["End_Timestamp"]) - s) / 1e3:6.1f} {r["Function"]} tid {r["Thread_Id"]}')
PY
rm -rf $O/tr
