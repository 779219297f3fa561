"""HBM traffic per launch shape from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out/fetch -o b -- python bench.py --steps 3 --warmup 1 \
        --no-settle --no-cpu-baseline --roofline-kernel none --dump-launch-order out/order.json
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out/write -o b -- python bench.py ... (same)
    python tools/pmc_traffic.py out/fetch/b_counter_collection.csv out/write/b_counter_collection.csv out/order.json \
        profiles/rNN_pmc_hbm_traffic.csv profiles/rNN_traffic.json

Every GEMM entry point launches exactly one kernel of its family (linear_nt_kernel for forward / dX, linear_dw_kernel for
dW, linear_bwd_fused_kernel for the fused dX+dW pass; the partial-tile reductions that follow are separate small kernels and
not counted here), in the same order every step, so the k-th family launch of a step belongs to the k-th such call of bench.py's launch
order.  hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: units are KiB and gfx950 reports half of a wide coalesced read
(MI355X_MICROARCH.md, HBM section)."""
import csv, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudlib_amd.buildinfo import csrc_sha  # noqa: E402

FAMILY = {"linear_nt_kernel": ("pcl_linear_fwd_rows_f32", "pcl_linear_fwd_gmax_f32", "pcl_linear_fwd_f32", "pcl_linear_bwd_dx_rows_f32"),
          "linear_fwd_res_kernel": (),
          "linear_dw_kernel": ("pcl_linear_bwd_dw_rows_f32",),
          "linear_bwd_fused_kernel": ("pcl_linear_bwd_fused_rows_f32",)}
# forward launches of hidden set-abstraction layers (>= 32768 rows in this workload) take the resident-weight kernel (round 3)
RES_TAGS = {"fwd64x64", "fwd64x128", "fwd128x128", "fwd128x256"}


def family_of(name, tag, fam=""):
    if fam:                      # round 4: the launch order records the kernel each entry point chose (pcl_last_launch_kernel)
        return fam
    if name.startswith("pcl_linear_fwd") and tag in RES_TAGS:
        return "linear_fwd_res_kernel"
    for fam, entries in FAMILY.items():
        if name in entries:
            return fam
    return None


def family_series(path, counter, families=FAMILY):
    out = defaultdict(list)                                  # family -> [value per dispatch, in dispatch order]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        if r["Counter_Name"] != counter:
            continue
        for fam in families:
            if f"pcl::{fam}<" in r["Kernel_Name"]:          # not group_linear_dw_kernel & co.
                out[fam].append(float(r["Counter_Value"]))
    return out


def main():
    fetch_csv, write_csv, order_json, out_csv, out_json = sys.argv[1:6]
    workload = sys.argv[6] if len(sys.argv) > 6 else "PointNet++ SSG cls B=32 N=1024, duplicate-compacted padded rows"
    command = sys.argv[7] if len(sys.argv) > 7 else "python bench.py --steps 3 --warmup 1 --no-settle"
    order = [tuple(o) + ("",) * (3 - len(o)) for o in json.load(open(order_json))["step_launch_order"]]
    families = sorted({family_of(*o) for o in order} - {None, ""})
    fetch, write = family_series(fetch_csv, "FETCH_SIZE", families), family_series(write_csv, "WRITE_SIZE", families)
    per_tag, lines = {}, []
    for fam in families:
        calls = [(n, t) for n, t, k in order if family_of(n, t, k) == fam]
        n = len(calls)
        f, w = fetch[fam], write[fam]
        if not (n and f and w and len(f) % n == 0 and len(w) % n == 0):
            print(f"# {fam}: {n} calls per step do not divide {len(f)} / {len(w)} dispatches -- skipped", file=sys.stderr)
            continue
        for k, (name, tag) in enumerate(calls):
            fk = f[k::n]; wk = w[k::n]
            fm, wm = sum(fk) / len(fk), sum(wk) / len(wk)
            b = (2 * fm + wm) * 1024
            per_tag[f"{name}:{tag}"] = round(b)
            lines.append((name, tag, len(fk), fm, wm, b))
    with open(out_csv, "w") as fh:
        fh.write(f"# {workload}\n# two separate passes: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- {command}\n")
        fh.write("# per-launch means; FETCH_SIZE/WRITE_SIZE in KiB; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction, MI355X_MICROARCH.md HBM section)\n")
        fh.write("entry_point,shape,launches,FETCH_SIZE,WRITE_SIZE,hbm_bytes\n")
        for l in lines:
            fh.write(f"{l[0]},{l[1]},{l[2]},{l[3]:.1f},{l[4]:.1f},{l[5]:.0f}\n")
    json.dump({"source": out_csv, "workload": workload,
               "csrc_sha": csrc_sha(),          # bench.py prints these numbers only for the same kernel sources
               "per_launch_hbm_bytes": per_tag}, open(out_json, "w"), indent=1)
    for l in lines:
        print(f"{l[0]:30s} {l[1]:14s} {l[5] / 1e6:8.1f} MB")


if __name__ == "__main__":
    main()
