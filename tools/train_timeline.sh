#!/bin/bash
# kernels of the LAST training iteration (bs = 4, nuScenes pillar model) in time order with their queues, and how the two streams
# (data-gradient chain / weight gradients) share the iteration:  tools/train_timeline.sh [extra env as VAR=VALUE ...]
# writes gpurun_out/train_tl/timeline.txt and prints the summary
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/train_tl
rm -rf "$OUT"; mkdir -p "$OUT"
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/raw" -o s -- python3 "$ROOT/bench.py" --mode train --steps 4 --warmup 2 > "$OUT/run.log" 2> "$OUT/run.err" || { tail -5 "$OUT/run.err"; exit 1; }
tail -1 "$OUT/run.log" | cut -c1-200
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/raw/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
ends = [i for i, r in enumerate(rows) if "adam_step_kernel" in r["Kernel_Name"]]
a, b = ends[-2] + 1, ends[-1]
it = rows[a:b + 1]
t0 = int(it[0]["Start_Timestamp"])
span = int(it[-1]["End_Timestamp"]) - t0
queues = {}
for r in it:
    queues.setdefault(r["Queue_Id"], []).append((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, name(r)))
main = max(queues, key=lambda q: len(queues[q]))
with open(out + "/timeline.txt", "w") as fh:
    for r in it:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        fh.write(f"{s / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  {'main' if r['Queue_Id'] == main else 'q' + r['Queue_Id']:>5}  {name(r)[:80]}\n")


def union(iv):
    iv = sorted(iv)
    res, cs, ce = 0, None, None
    merged = []
    for s, e, *_ in iv:
        if cs is None or s > ce:
            if cs is not None:
                merged.append((cs, ce))
            cs, ce = s, e
        else:
            ce = max(ce, e)
    if cs is not None:
        merged.append((cs, ce))
    return merged


def overlap(a_, b_):
    i = j = 0
    tot = 0
    while i < len(a_) and j < len(b_):
        s, e = max(a_[i][0], b_[j][0]), min(a_[i][1], b_[j][1])
        if e > s:
            tot += e - s
        if a_[i][1] < b_[j][1]:
            i += 1
        else:
            j += 1
    return tot


print(f"iteration span {span / 1e6:.3f} ms, {len(it)} kernels, queues: " + ", ".join(f"{'main' if q == main else q}: {len(v)}" for q, v in queues.items()))
um = union(queues[main])
side = [x for q, v in queues.items() if q != main for x in v]
us = union(side)
bm, bs = sum(e - s for s, e in um), sum(e - s for s, e in us)
both = overlap(um, us)
print(f"main busy {bm / 1e6:.3f} ms, side busy {bs / 1e6:.3f} ms, both {both / 1e6:.3f} ms, main only {(bm - both) / 1e6:.3f}, side only {(bs - both) / 1e6:.3f}, "
      f"neither {(span - bm - bs + both) / 1e6:.3f}")
# the main queue's kernels: time alone and time beside a side-stream kernel, by kernel name
agg = {}
for s, e, n in queues[main]:
    ov = overlap([(s, e)], us)
    d = agg.setdefault(n, [0, 0, 0])
    d[0] += 1; d[1] += e - s; d[2] += ov
print("main-queue kernels (calls, ms, of which beside the side stream):")
for n, d in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"  {n[:60]:60s} {d[0]:4d} {d[1] / 1e6:7.3f} {d[2] / 1e6:7.3f}")
agg = {}
for s, e, n in side:
    d = agg.setdefault(n, [0, 0])
    d[0] += 1; d[1] += e - s
print("side-queue kernels (calls, ms):")
for n, d in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]:
    print(f"  {n[:60]:60s} {d[0]:4d} {d[1] / 1e6:7.3f}")
# gaps of the main queue longer than 5 us
gaps = [(um[i + 1][0] - um[i][1], um[i][1]) for i in range(len(um) - 1)]
print(f"main-queue gaps: {sum(g for g, _ in gaps) / 1e6:.3f} ms in {len(gaps)}; longer than 20 us:")
for g, at in sorted(gaps, reverse=True)[:12]:
    if g > 20000:
        print(f"  {g / 1e3:7.1f} us at {at / 1e3:9.1f}")
PY
[ -n "$KEEP_RAW" ] || rm -rf "$OUT/raw"
