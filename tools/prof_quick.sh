#!/bin/bash
# kernel stats of the default bench workload (quick look), one frame in flight or STREAMS=4:  [STREAMS=4] bash tools/prof_quick.sh [pattern]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/quick
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o s -- python3 $ROOT/bench.py --steps 40 --warmup 5 --streams ${STREAMS:-1} --no-cpu-baseline --no-train-leg --no-roofline-events --no-batched --no-c4 --no-c5 > "$OUT/run.log" 2> "$OUT/run.err" || { tail -5 "$OUT/run.err"; exit 1; }
python3 - "$OUT" "${1:-.}" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
frames = next((float(r['Calls']) for r in rows if 'fused_polar_index_kernel' in r['Name']), 46.0)      # one launch per frame
tot = 0
for r in rows[:40]:
    name = r['Name'].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    per = float(r['Calls']) / frames
    tot += float(r['TotalDurationNs']) / frames
    if re.search(sys.argv[2], name):
        print(f"{name[:70]:70s} {per:6.2f}/frame {float(r['AverageNs'])/1e3:9.2f} us  {float(r['TotalDurationNs'])/frames/1e3:8.1f} us/frame")
print("sum of the top 40 kernels per frame: %.1f us" % (tot / 1e3))
PY
