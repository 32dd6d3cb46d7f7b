#!/bin/bash
# the sparse encoder's launches of ONE step of the Waymo PARTNER config (bs 2), in launch order with their durations: tools/c4_sparse_timeline.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c4_tl
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o s -- python3 "$ROOT/tools/c4_e2e_profile.py" f32 3 2 > "$OUT/run.log" 2> "$OUT/run.err" || { tail -5 "$OUT/run.err"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "index_from_coords" in r["Kernel_Name"] or "coords_bitmap" in r["Kernel_Name"]]
names = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
# last step: from the last sparse-index build to the first to_dense kernel after it
starts = [i for i, r in enumerate(rows) if "mark_coords" in r["Kernel_Name"] or "index_from" in r["Kernel_Name"] or "permute_rows_kernel" in r["Kernel_Name"]]
a = [i for i, r in enumerate(rows) if "permute_rows_kernel" in r["Kernel_Name"]][-1]
b = [i for i, r in enumerate(rows) if "to_dense_kernel" in r["Kernel_Name"] and i > a][0]
t0 = int(rows[a]["Start_Timestamp"])
tot = 0
for r in rows[a - 12:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = names(r)
    if any(k in n for k in ("sparse_conv", "group_rows", "neighbor", "mark_down", "emit", "tile_", "permute", "to_dense", "hard_select", "zero_fill")):
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):6d} x {r['Grid_Size_Y']}  {n[:60]}  q{r.get('Queue_Id','')}")
print("span permute -> to_dense: %.1f us" % ((int(rows[b]["End_Timestamp"]) - t0) / 1e3))
PY
