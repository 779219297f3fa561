"""The round-5 ISA lens as a tool: for every kernel of a HIP source whose mangled name contains one of the given
substrings, read hipcc's gfx950 assembly and report what made three "tuned" kernels run at half speed in round 5:

  * loads the wave WAITS for right behind their issue: `s_waitcnt vmcnt(0)` within a few instructions of a buffer / global load with
    no other work between (the compiler sank a prefetch to its first use, or a touched value forces the wait);
  * waits that ALSO cover stores: a `vmcnt(n)` small enough to drain stores that were issued after the newest load still needed
    (loads and stores retire through one counter);
  * `s_cbranch_execz` directly in front of a store (a store under a per-lane branch);
  * scratch (spilled registers / arrays in private memory) and the register count.

    python tools/isa_lens.py pointcloudlib_amd/csrc/mlp.hip linear_fwd_res_kernel [more substrings]
"""
import os
import re
import subprocess
import sys
import tempfile

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics", "-Wno-unused-function", "-S", "--cuda-device-only"]
LOAD = re.compile(r"^\s*(buffer_load|global_load|flat_load|scratch_load)")
STORE = re.compile(r"^\s*(buffer_store|global_store|flat_store|scratch_store)")
WAIT = re.compile(r"^\s*s_waitcnt\s+(.*)")
MFMA = re.compile(r"^\s*v_mfma")
VALU = re.compile(r"^\s*v_")
LDS = re.compile(r"^\s*ds_")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def analyse(body):
    ins = [l.split(";")[0].rstrip() for l in body if l.strip() and not l.strip().startswith((";", ".", "//")) and not re.match(r"^\S+:$", l.strip())]
    n_load = sum(1 for l in ins if LOAD.match(l) and "scratch" not in l)
    n_store = sum(1 for l in ins if STORE.match(l) and "scratch" not in l)
    n_mfma = sum(1 for l in ins if MFMA.match(l))
    n_valu = sum(1 for l in ins if VALU.match(l)) - n_mfma
    n_lds = sum(1 for l in ins if LDS.match(l))
    n_scr = sum(1 for l in ins if "scratch_" in l)
    wait0_behind_load = 0            # vmcnt(0) with a load <= 3 instructions earlier and no MFMA / LDS / >2 VALU between
    wait_covers_store = 0            # a vmcnt wait with un-retired stores issued since the last wait, small enough to drain them
    execz_store = 0
    since_load, work_since_load = None, 0
    stores_since_wait = 0
    loads_since_wait = 0
    for i, l in enumerate(ins):
        if LOAD.match(l) and "scratch" not in l:
            since_load, work_since_load = 0, 0
            loads_since_wait += 1
            continue
        if STORE.match(l) and "scratch" not in l:
            stores_since_wait += 1
            if i and "s_cbranch_execz" in ins[i - 1] or (i > 1 and "s_cbranch_execz" in ins[i - 2]):
                execz_store += 1
        m = WAIT.match(l)
        if m and "vmcnt" in m.group(1):
            n = int(re.search(r"vmcnt\((\d+)\)", m.group(1)).group(1))
            if n == 0 and since_load is not None and since_load <= 3 and work_since_load <= 2:
                wait0_behind_load += 1
            if stores_since_wait and n < stores_since_wait + 0 and loads_since_wait:
                wait_covers_store += 1
            stores_since_wait = min(stores_since_wait, n)
            loads_since_wait = 0
        if since_load is not None:
            since_load += 1
            if MFMA.match(l) or LDS.match(l) or VALU.match(l):
                work_since_load += 1
    return dict(instructions=len(ins), loads=n_load, stores=n_store, mfma=n_mfma, valu=n_valu, lds=n_lds, scratch_ops=n_scr,
                wait0_right_behind_a_load=wait0_behind_load, vmcnt_waits_that_drain_stores=wait_covers_store, execz_in_front_of_a_store=execz_store)


def main():
    src, pats = sys.argv[1], sys.argv[2:]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-I", os.path.dirname(src), "-o", out, src], stderr=subprocess.DEVNULL)
        txt = open(out).read()
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", txt):
        meta[m.group(1)] = (int(m.group(2)), int(m.group(3)), int(m.group(4)))
    names = [n for n in re.findall(r"^(_Z\S+):\s*(?:;.*)?$", txt, re.M) if any(p in n for p in pats)]
    dm = demangle(names)
    for n in names:
        i = txt.index("\n" + n + ":")
        j = txt.index(".Lfunc_end", i)
        r = analyse(txt[i:j].split("\n")[2:])
        scr, vg, sp = meta.get(n, (None, None, None))
        print(f"{dm[n][:150]}\n    vgprs {vg}  spilled {sp}  scratch {scr} B | " + "  ".join(f"{k} {v}" for k, v in r.items()))


if __name__ == "__main__":
    main()
