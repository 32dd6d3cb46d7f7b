set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
MODE=${1:-f32}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/c4_e2e_$MODE
mkdir -p "$OUT"
python3 "$ROOT/tools/c4_e2e_profile.py" $MODE 20
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o s -- python3 "$ROOT/tools/c4_e2e_profile.py" $MODE 10 > "$OUT/run.log" 2> "$OUT/run.err" || { tail -5 "$OUT/run.err"; exit 1; }
cat "$OUT/run.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6/13:.3f} ms/iter over {len(rows)} kernels")
for r in rows[:45]:
    print(f"{r['Name'][:110]:110s} {float(r['Calls'])/13:7.1f}/it {float(r['AverageNs'])/1e3:9.2f} us {float(r['TotalDurationNs'])/1e6/13:8.3f} ms/it {r['Percentage']:>6s} %")
PY
