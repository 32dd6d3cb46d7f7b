#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats output dir -> the small tracked CSV under profiles/ (per-kernel calls / total / average), with
the per-step normalisation written in the header.   python tools/kernel_stats_summary.py <stats_dir> <out.csv> <steps> [note]
With a PMC dir (SQ counters):                       python tools/kernel_stats_summary.py --mfma <pmc_dir> <out.csv>"""
import collections
import csv
import glob
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def stats(stats_dir, out, steps, note=""):
    f = glob.glob(stats_dir + "/**/*kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out, "w") as o:
        o.write(f"# {note}; {steps} steps profiled; total kernel time {tot / 1e6 / steps:.3f} ms per step\n")
        o.write("kernel,calls_per_step,avg_us,ms_per_step,percent,min_ns,max_ns\n")
        for r in rows:
            o.write(f"\"{short(r['Name'])}\",{float(r['Calls']) / steps:.2f},{float(r['AverageNs']) / 1e3:.2f},{float(r['TotalDurationNs']) / 1e6 / steps:.4f},"
                    f"{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")
    print(f"{out}: {tot / 1e6 / steps:.3f} ms of kernels per step")
    for r in rows[:25]:
        print(f"  {short(r['Name'])[:90]:90s} {float(r['Calls']) / steps:7.1f}/step {float(r['AverageNs']) / 1e3:9.2f} us {float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms/step")


def mfma(pmc_dir, out):
    p = glob.glob(pmc_dir + "/**/*_counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(out, "w") as f:
        f.write("kernel,launches,SQ_VALU_MFMA_BUSY_CYCLES_avg,GRBM_GUI_ACTIVE_avg(sum of 8 XCDs),mfma_util,SQ_WAIT_ANY/SQ_WAVE_CYCLES,"
                "SQ_WAIT_INST_ANY/SQ_WAVE_CYCLES,SQ_LDS_BANK_CONFLICT_avg\n")
        for k in sorted(acc, key=lambda k: -sum(acc[k].get("GRBM_GUI_ACTIVE", [0]))):
            a = lambda c: (sum(acc[k][c]) / len(acc[k][c])) if acc[k].get(c) else float("nan")   # noqa: E731
            busy, gui, wc = a("SQ_VALU_MFMA_BUSY_CYCLES"), a("GRBM_GUI_ACTIVE"), a("SQ_WAVE_CYCLES")
            if busy != busy or not busy:
                continue
            f.write(f"\"{k}\",{len(acc[k]['SQ_WAVE_CYCLES'])},{busy:.0f},{gui:.0f},{busy / (gui / 8 * 1024):.3f},{a('SQ_WAIT_ANY') / wc:.3f},"
                    f"{a('SQ_WAIT_INST_ANY') / wc:.3f},{a('SQ_LDS_BANK_CONFLICT'):.0f}\n")
    print(open(out).read())


if __name__ == "__main__":
    if sys.argv[1] == "--mfma":
        mfma(sys.argv[2], sys.argv[3])
    else:
        stats(sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else "")
