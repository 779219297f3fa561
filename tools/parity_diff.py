"""Parity rows that moved by more than 10 % between two rounds' reports (a row that drifts toward its cap must be seen
the round it moves, not the round it fails).
    python tools/parity_diff.py profiles/r05_parity_reports.txt profiles/r06_parity_reports.txt [threshold = 0.10]
Feature rows are compared by their distance from fp64 in units of the 1e-5 bound, WAIVER lines by value against cap, gradient rows
(only those the reports print: the not-ok ones and the worst few) by relative L2."""
import re
import sys


def rows(path):
    out, title = {}, "?"
    for line in open(path, errors="replace"):
        m = re.match(r"\[parity (.*)\]", line.strip())
        if m:
            title = m.group(1)
            continue
        m = re.match(r"\s+(.*?)\s+max\|hip-fp64\| \S+ \(\s*([\d.]+) x bound\)", line)
        if m:
            out[(title, "feature", m.group(1).strip())] = float(m.group(2))
            continue
        m = re.match(r"\s+WAIVER (\S+ .*?): (.*?): ([\d.e+-]+) \(cap ([\d.e+-]+)\)", line)
        if m:
            out[(title, "waiver " + m.group(1), m.group(2).strip())] = (float(m.group(3)), float(m.group(4)))
            continue
        m = re.match(r"\s+grad (\S+)\s+relL2 hip ([\d.e+-]+) / fp32-restatement ([\d.e+-]+)", line)
        if m:
            out[(title, "grad", m.group(1))] = float(m.group(2))
    return out


def main():
    a, b = rows(sys.argv[1]), rows(sys.argv[2])
    thr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.10
    n = 0
    print(f"# parity rows that moved by more than {thr:.0%}: {sys.argv[1]} -> {sys.argv[2]}")
    for k in sorted(set(a) & set(b)):
        va, vb = a[k], b[k]
        cap = ""
        if isinstance(va, tuple):
            cap = f"   cap {vb[1]:g} ({vb[0] / vb[1]:.0%} of it)"
            va, vb = va[0], vb[0]
        if k[1] == "grad" and max(va, vb) < 1e-4:
            continue
        if va > 0 and abs(vb - va) / va > thr:
            n += 1
            print(f"  {'WORSE ' if vb > va else 'better'}  {k[0]} | {k[1]} | {k[2]}: {va:.4g} -> {vb:.4g} ({(vb - va) / va:+.0%}){cap}")
    only = sorted(set(b) - set(a))
    if only:
        print(f"# rows only in the newer report: {len(only)} (e.g. {only[0][0]} | {only[0][2]})")
    print(f"# {n} rows moved; {len(set(a) & set(b))} compared")


if __name__ == "__main__":
    main()
