#!/bin/bash
# rocprofv3 kernel trace + stats of a bench.py run (the program goes directly after `--`).
#   tools/prof_stats.sh <out_dir_under_gpurun_out> <bench args...>
# prints the kernel stats table; summaries to commit are made by tools/summarize_profile.py
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1
shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o s -- python3 "$ROOT/bench.py" "$@" > "$OUT/bench.log" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:60]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.2f} us {r['Percentage']:>6s} %")
PY
