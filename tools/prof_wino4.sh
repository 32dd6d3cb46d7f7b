#!/bin/bash
# kernel-only durations of tools/wino4_bench.py under rocprofv3:  bash tools/prof_wino4.sh [shapes...]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/wino4
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o s -- python3 "$ROOT/tools/wino4_bench.py" "$@" > "$OUT/run.log" 2> "$OUT/run.err" || { tail -5 "$OUT/run.err"; exit 1; }
cat "$OUT/run.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
d = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name']
    if 'conv_wino4' in n:
        key = (n.split('(')[0][-40:], r['Grid_Size_X'], r.get('LDS_Block_Size', ''))
        d.setdefault(key, [])
        d[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
# launches of different shapes share a kernel name: print runs of 35 launches in order
seq = [((r['Kernel_Name'].split('(')[0][-30:], r['Grid_Size_X']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows if 'conv_wino4' in r['Kernel_Name']]
for i in range(0, len(seq), 300):
    v = sorted(t for _, t in seq[i + 100:i + 300])
    print(seq[i][0], "median %.1f us  min %.1f" % (v[len(v) // 2], v[0]))
PY
