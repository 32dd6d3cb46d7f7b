#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 rocpd database (the default output format of ROCm 7.2):
python tools/rocpd_summary.py <results.db> [marker-kernel-substring] [skip-steps] [out.csv]
Steps are delimited by the marker kernel (default: adam_step); the first `skip-steps` are dropped."""
import collections
import sqlite3
import sys


def main(db_path, marker="adam_step", skip=3, out_csv=None):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    skip = int(skip)
    if len(marks) > skip:
        sel, steps = rows[marks[skip - 1] + 1:marks[-1] + 1], len(marks) - skip
    else:
        sel, steps = rows, 1
    acc = collections.defaultdict(lambda: [0, 0])
    for n, s, e in sel:
        n = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[n][0] += 1
        acc[n][1] += e - s
    tot = sum(v[1] for v in acc.values())
    wall = sel[-1][2] - sel[0][1]
    lines = [f"# steps {steps}  kernel_ms_per_step {tot / 1e6 / steps:.3f}  wall_ms_per_step {wall / 1e6 / steps:.3f}",
             "kernel,calls_per_step,ms_per_step,avg_us,percent"]
    for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"\"{n}\",{c / steps:.1f},{t / 1e6 / steps:.4f},{t / c / 1e3:.1f},{100.0 * t / tot:.2f}")
    text = "\n".join(lines)
    if out_csv:
        open(out_csv, "w").write(text + "\n")
    print("\n".join(lines[:40]))


if __name__ == "__main__":
    main(*sys.argv[1:5])
