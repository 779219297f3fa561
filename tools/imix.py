"""Instruction mix of a kernel's K-step body and of the rest (prologue + epilogues) from hipcc's assembly.
    python tools/imix.py pointcloudlib_amd/csrc/mlp.hip <mangled-name-substring> ...
VALU instructions matter most: fp32-input MFMA shares the vector ALU datapath on gfx950 (tools/ubench/coissue.hip)."""
import subprocess, sys, tempfile, os
src, subs = sys.argv[1], sys.argv[2:]
out = os.path.join(tempfile.gettempdir(), "imix.s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
                "-S", "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL, check=True)
s = open(out).read()
def count(lines):
    c = {"mfma": 0, "valu": 0, "salu": 0, "vmem": 0, "lds": 0}
    for l in lines:
        l = l.strip()
        if not l or l[0] in ";." or l.endswith(":"):
            continue
        op = l.split()[0]
        k = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else
             "vmem" if op.startswith(("global_", "buffer_", "scratch_", "flat_")) else "lds" if op.startswith("ds_") else None)
        if k:
            c[k] += 1
    return c
import re
for sub in subs:
    for m in re.finditer(r"\n(_Z\w*" + re.escape(sub) + r"\w*): ", s):
        name = m.group(1)
        L = s[m.end():s.index("s_endpgm", m.end())].split("\n")
        bar = [i for i, l in enumerate(L) if "s_barrier" in l]
        mf = [i for i, l in enumerate(L) if "v_mfma" in l]
        if not mf:
            continue
        before = [i for i in bar if i < mf[0]]
        start = before[-2] if len(before) >= 2 else (before[0] if before else 0)
        end = mf[0]
        for a, b in zip(mf, mf[1:]):
            if b - a > 40:
                break
            end = b
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        print(dem[:90], "\n    K-step body (inner 8-k block counted once):", count(L[start:end + 3]), "\n    rest:", count(L[:start] + L[end + 3:]))
