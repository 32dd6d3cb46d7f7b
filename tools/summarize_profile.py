#!/usr/bin/env python3
"""Condense rocprofv3 outputs (kernel stats + PMC passes) into the small tracked summaries under
profiles/.   python tools/summarize_profile.py <stats_dir> <pmc_fetch_dir> <pmc_write_dir> <bench_log> <out_prefix>"""
import collections
import csv
import glob
import json
import os
import sys


def first(pattern):
    g = glob.glob(pattern, recursive=True)
    return g[0] if g else None


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main(stats_dir, fetch_dir, write_dir, bench_log, out_prefix):
    os.makedirs(os.path.dirname(out_prefix), exist_ok=True)
    stats = list(csv.DictReader(open(first(stats_dir + "/**/*_kernel_stats.csv"))))
    with open(out_prefix + "_kernel_stats.csv", "w") as f:
        f.write("kernel,calls,total_ns,avg_ns,percent,min_ns,max_ns\n")
        for r in stats:
            f.write(f"\"{short(r['Name'])}\",{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},{r['Percentage']},"
                    f"{r['MinNs']},{r['MaxNs']}\n")
    bench = None
    for line in open(bench_log):
        if line.startswith("{"):
            bench = json.loads(line)
    # PMC: per kernel name, average per launch; FETCH_SIZE/WRITE_SIZE are in KB
    def pmc(d):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        p = first(d + "/**/*_counter_collection.csv")
        if not p:
            return acc
        for r in csv.DictReader(open(p)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        return acc
    fe, wr = pmc(fetch_dir), pmc(write_dir)
    rows = []
    for k in sorted(set(fe) | set(wr)):
        n = max(len(v) for v in list(fe[k].values()) + list(wr[k].values()))
        g = lambda a, c: (sum(a[k][c]) / len(a[k][c])) if a[k].get(c) else float("nan")
        fetch_kb, write_kb = g(fe, "FETCH_SIZE"), g(wr, "WRITE_SIZE")
        hit, miss = g(fe, "TCC_HIT_sum"), g(wr, "TCC_MISS_sum")
        # gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section)
        rows.append((k, n, fetch_kb, 2 * fetch_kb, write_kb, hit, miss))
    with open(out_prefix + "_pmc_traffic.csv", "w") as f:
        f.write("kernel,launches,FETCH_SIZE_KB_raw_avg,fetch_KB_corrected_x2_avg,WRITE_SIZE_KB_avg,TCC_HIT_avg,TCC_MISS_avg\n")
        for r in rows:
            f.write(f"\"{r[0]}\",{r[1]},{r[2]:.1f},{r[3]:.1f},{r[4]:.1f},{r[5]:.0f},{r[6]:.0f}\n")
    # optional SQ pass: MFMA utilisation = MFMA busy cycles / (GPU-active cycles per XCD * 1024 SIMDs)
    sq_dir = os.environ.get("SQ_PMC_DIR")
    if sq_dir:
        sq = pmc(sq_dir)
        with open(out_prefix + "_pmc_mfma.csv", "w") as f:
            # mfma_util divides by the cycles the GPU is active AROUND the kernel in the counter pass, which carry a fixed ~11-14 us of
            # counter start / stop per launch (a 19 us kernel shows 30 us of GRBM_GUI_ACTIVE): the last two columns put the same busy cycles
            # over the kernel's duration in the undisturbed kernel-trace pass, at the 2.4 GHz peak clock (the clock under these kernels is
            # 2.25-2.4 GHz, tools/micro/wchain_check.hip)
            trace_ns = {short(r["Name"]): float(r["AverageNs"]) for r in stats}
            f.write("kernel,launches,SQ_VALU_MFMA_BUSY_CYCLES_avg,GRBM_GUI_ACTIVE_avg(sum of 8 XCDs),mfma_util,SQ_WAIT_ANY/SQ_WAVE_CYCLES,"
                    "SQ_WAIT_INST_ANY/SQ_WAVE_CYCLES,SQ_LDS_BANK_CONFLICT_avg,avg_us_kernel_trace,mfma_busy_over_trace_time_at_2.4GHz\n")
            for k in sorted(sq):
                a = lambda c: (sum(sq[k][c]) / len(sq[k][c])) if sq[k].get(c) else float("nan")
                busy, gui, wc = a("SQ_VALU_MFMA_BUSY_CYCLES"), a("GRBM_GUI_ACTIVE"), a("SQ_WAVE_CYCLES")
                if not busy or busy != busy or busy == 0:
                    continue
                t = trace_ns.get(k)
                f.write(f"\"{k}\",{len(sq[k]['SQ_WAVE_CYCLES'])},{busy:.0f},{gui:.0f},{busy / (gui / 8 * 1024):.3f},"
                        f"{a('SQ_WAIT_ANY') / wc:.3f},{a('SQ_WAIT_INST_ANY') / wc:.3f},{a('SQ_LDS_BANK_CONFLICT'):.0f},"
                        + (f"{t / 1e3:.2f},{busy / 1024 / (t * 2.4):.3f}" if t else ",") + "\n")
    if bench:
        json.dump(bench, open(out_prefix + "_bench_line.json", "w"), indent=1)
    print("wrote", out_prefix + "_kernel_stats.csv", out_prefix + "_pmc_traffic.csv")


if __name__ == "__main__":
    main(*sys.argv[1:6])
