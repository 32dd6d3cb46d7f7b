#!/bin/bash
# kernel times of tools/wino_bench.py under rocprofv3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/wino
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o s -- python3 "$ROOT/tools/wino_bench.py" > "$OUT/run.log" 2> "$OUT/run.err" || { tail -5 "$OUT/run.err"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
d = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if 'conv_wino' in n or 'conv_mfma' in n:
        key = (n.split('(')[0][-60:], r['Grid_Size_X'], r['Grid_Size_Y'])
        d[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in d.items():
    v = sorted(v)
    print(k, len(v), "median %.1f us" % v[len(v) // 2])
PY
