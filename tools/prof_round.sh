#!/bin/bash
# Profiles of one round, on the GPU box:  tools/prof_round.sh r2
#   1. kernel trace + stats, one frame in flight (the regime the roofline object is quoted in)
#   2. kernel trace + stats of the DEFAULT bench command (frames in flight: bench.py's --streams default)
#   3. PMC passes (separate runs, --kernel-trace only next to --pmc): FETCH_SIZE, WRITE_SIZE, SQ counters
# and the condensed summaries under profiles/<prefix>_* (tools/summarize_profile.py).  The program goes directly after `--`.
set -e
PFX=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$PFX
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py"
ONE="--steps 40 --warmup 5 --streams 1 --no-cpu-baseline --no-train-leg --no-roofline-events --no-batched --no-c4 --no-c5"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats1" -o s -- python3 $B $ONE > "$OUT/bench_1stream.log" 2> "$OUT/bench_1stream.err"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats4" -o s -- python3 $B --steps 40 --warmup 5 --no-cpu-baseline --no-train-leg --no-roofline-events --no-batched --no-c4 --no-c5 > "$OUT/bench_4streams.log" 2> "$OUT/bench_4streams.err"
# r6: the counter passes run the DEFAULT regime (frames in flight: the kernel forms `value` is timed with -- conv_wchain3_kernel among them);
# a counter pass serialises the dispatches, so every kernel is still measured alone.  bench.py's roofline.traffic reads the summary.
PM="--steps 12 --warmup 3 --no-cpu-baseline --no-train-leg --no-roofline-events --no-batched --no-c4 --no-c5 --no-sustained"
timeout 900 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d "$OUT/fetch" -o f -- python3 $B $PM > /dev/null 2> "$OUT/pmc_fetch.err"
timeout 900 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/write" -o w -- python3 $B $PM > /dev/null 2> "$OUT/pmc_write.err"
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$OUT/sq" -o q -- python3 $B $PM > /dev/null 2> "$OUT/pmc_sq.err" || echo "SQ pass failed (see pmc_sq.err)"
cd "$ROOT"
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
SQ_PMC_DIR="$OUT/sq" python3 tools/summarize_profile.py "$OUT/stats1" "$OUT/fetch" "$OUT/write" "$OUT/bench_default.json" "$OUT/sum/${PFX}"
python3 - "$OUT" "$PFX" <<'PY'
import csv, glob, sys
out, pfx = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/stats4/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open(f"{out}/sum/{pfx}_inflight_kernel_stats.csv", "w") as o:
    o.write("kernel,calls,total_ns,avg_ns,percent,min_ns,max_ns\n")
    for r in rows:
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        o.write(f"\"{name}\",{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")
print("summaries in", out + "/sum")
PY
ls "$OUT/sum"
[ -n "$KEEP_RAW" ] || rm -rf "$OUT/stats1" "$OUT/stats4" "$OUT/fetch" "$OUT/write" "$OUT/sq"   # gpurun copies back at most 64 MiB
