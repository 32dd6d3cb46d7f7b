#!/bin/bash
# kernel statistics of the nuScenes training iteration (bs = 4):  bash tools/prof_train.sh  ->  gpurun_out/prof_train/sum.csv
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_train
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o s -- python3 $ROOT/bench.py --mode train --steps 6 --warmup 2 > "$OUT/run.log" 2> "$OUT/run.err" || { tail -5 "$OUT/run.err"; exit 1; }
tail -1 "$OUT/run.log" | cut -c1-200
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
n = 8.0
with open(sys.argv[1] + "/sum.csv", "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --mode train --steps 6 --warmup 2   (8 iterations of bs = 4; per-iteration = total / 8)\n")
    o.write("kernel,calls,total_ns,avg_ns,percent\n")
    for r in rows:
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        o.write(f"\"{name}\",{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},{r['Percentage']}\n")
tot = sum(float(r['TotalDurationNs']) for r in rows) / n / 1e6
print("kernel time per iteration %.2f ms" % tot)
for r in rows[:28]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print(f"{name[:64]:64s} {float(r['Calls'])/n:6.1f}/it {float(r['AverageNs'])/1e3:9.1f} us {float(r['TotalDurationNs'])/n/1e6:7.3f} ms/it")
PY
