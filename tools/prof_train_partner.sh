set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/train_partner
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o s -- python3 "$ROOT/tools/train_partner_profile.py" --steps 3 > "$OUT/run.log" 2> "$OUT/run.err" || { tail -5 "$OUT/run.err"; exit 1; }
cat "$OUT/run.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:40]:
    print(f"{r['Name'][:110]:110s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.2f} us {float(r['TotalDurationNs'])/1e6:9.2f} ms {r['Percentage']:>6s} %")
PY
