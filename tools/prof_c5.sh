#!/bin/bash
# rocprofv3 kernel stats of the C5 frame (RAW 10 sweeps -> boxes, eager launches): tools/prof_c5.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c5
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o s -- python3 "$ROOT/tools/c5_profile.py" > "$OUT/run.log" 2> "$OUT/run.err" || { tail -5 "$OUT/run.err"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows) / 13e3
print("kernel time per frame: %.1f us" % tot)
for r in rows[:22]:
    print(f"{r['Name'][:90]:90s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.2f} us {float(r['TotalDurationNs'])/13e3:9.1f} us/frame")
PY
