#!/bin/bash
# kernels of the LAST replay of the sparse encoder's hipGraph in time order (start, duration, queue): tools/c4_sparse_graph_timeline.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c4_gtl
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o s -- python3 "$ROOT/tools/c4_sparse_graph_timeline.py" > "$OUT/run.log" 2> "$OUT/run.err" || { tail -5 "$OUT/run.err"; exit 1; }
cat "$OUT/run.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
ends = [i for i, r in enumerate(rows) if "to_dense" in r["Kernel_Name"]]
b = ends[-1]
a = ends[-2] + 1
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  q{r.get('Queue_Id', '')}  {name(r)[:70]}")
print("replay span: %.1f us" % ((int(rows[b]["End_Timestamp"]) - t0) / 1e3))
PY
