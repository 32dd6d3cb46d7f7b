#!/bin/bash
# rocprofv3 kernel stats of the sparse encoder of the Waymo config on a 64-beam synthetic sweep: tools/prof_c4_sparse.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c4_sparse
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o s -- python3 "$ROOT/tools/c4_sparse_profile.py" > "$OUT/run.log" 2> "$OUT/run.err" || { tail -5 "$OUT/run.err"; exit 1; }
cat "$OUT/run.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:16]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.2f} us {float(r['TotalDurationNs'])/1e6:9.3f} ms {r['Percentage']:>6s} %")
PY
