#!/bin/bash
# rocprofv3 kernel stats of any python tool:  tools/prof_any.sh <tag> <steps profiled> tools/<script>.py [args...]
# -> gpurun_out/prof_<tag>/sum/<tag>_kernel_stats.csv  (the program goes directly after `--`)
TAG=$1; STEPS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT/sum"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python3 "$ROOT/$1" "${@:2}" > "$OUT/run.log" 2> "$OUT/run.err" || tail -5 "$OUT/run.err"
cat "$OUT/run.log"
cd "$ROOT"
python3 tools/kernel_stats_summary.py "$OUT/stats" "$OUT/sum/${TAG}_kernel_stats.csv" "$STEPS" "$*"
