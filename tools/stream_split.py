#!/usr/bin/env python3
"""rocprofv3 --kernel-trace output -> kernel time per HIP stream / queue (which stream is the critical path of an iteration?):
python tools/stream_split.py <trace_dir> [iterations]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
it = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(f)))
key = "Stream_Id" if "Stream_Id" in rows[0] else "Queue_Id"
per = collections.defaultdict(lambda: collections.defaultdict(float))
tot = collections.defaultdict(float)
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    per[r[key]][name] += d
    tot[r[key]] += d
t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
print(f"{key}: span {(t1 - t0) / 1e6:.1f} ms")
for s, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"  {key} {s}: {v / 1e3 / it:9.3f} ms per iteration")
    for n, d in sorted(per[s].items(), key=lambda kv: -kv[1])[:12]:
        print(f"      {n[:70]:70s} {d / 1e3 / it:8.3f}")
