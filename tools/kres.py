"""Kernel resource usage of one .hip source (VGPRs, spills, scratch, LDS, occupancy) from hipcc's remarks.
    python tools/kres.py pointcloudlib_amd/csrc/mlp.hip [substring]"""
import re, subprocess, sys
src = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                    "-munsafe-fp-atomics", *sys.argv[3:], "-c", "--cuda-device-only", "-o", "/dev/null", src,
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur, rows = None, {}
for l in r.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?): (\S+)", l)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = v; rows[cur] = {}
    elif cur:
        rows[cur][k] = v
names = subprocess.run(["c++filt"] + list(rows), capture_output=True, text=True).stdout.splitlines()
for n, (k, d) in zip(names, rows.items()):
    if sub in n:
        print(f"{n[:110]:110s} vgpr={d.get('VGPRs')} agpr={d.get('AGPRs')} spill={d.get('VGPRs Spill')} scratch={d.get('ScratchSize [bytes/lane]')} "
              f"occ={d.get('Occupancy [waves/SIMD]')} lds={d.get('LDS Size [bytes/block]')}")
