#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/wg_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d "$OUT" -o q -- python3 $ROOT/tools/wgrad_wino4_bench.py > "$OUT/run.log" 2> "$OUT/run.err" || { tail -5 "$OUT/run.err"; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r['Kernel_Name'].split('(')[0][-40:]
    if 'wgrad' in n:
        agg[(n, r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in agg.items():
    g = {n: sum(v) / len(v) for n, v in c.items()}
    busy = g.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(g.get('GRBM_GUI_ACTIVE', 1), 1) / 4 / 32   # per-XCD sum... rough
    print(k, {n: round(v) for n, v in g.items()})
PY
