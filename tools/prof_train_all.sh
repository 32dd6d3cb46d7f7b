#!/bin/bash
# Tracked kernel statistics of the two training iterations:  tools/prof_train_all.sh r3
#   <pfx>_train_kernel_stats.csv          python3 bench.py --mode train --steps 6 --warmup 2   (8 iterations of bs = 4, nuScenes pillar model)
#   <pfx>_train_partner_kernel_stats.csv  python3 tools/train_partner_profile.py --steps 3     (1 + 3 iterations of bs = 2, Waymo PARTNER detector)
# The program goes directly after `--`; raw traces are deleted (gpurun copies back at most 64 MiB).
PFX=${1:-r3}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_train_$PFX
rm -rf "$OUT"; mkdir -p "$OUT/sum"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train" -o s -- python3 $ROOT/bench.py --mode train --steps 6 --warmup 2 > "$OUT/train.log" 2> "$OUT/train.err" || tail -5 "$OUT/train.err"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/partner" -o s -- python3 $ROOT/tools/train_partner_profile.py --steps 3 > "$OUT/partner.log" 2> "$OUT/partner.err" || tail -5 "$OUT/partner.err"
tail -1 "$OUT/train.log" | cut -c1-300
cat "$OUT/partner.log"
cd "$ROOT"
python3 tools/kernel_stats_summary.py "$OUT/train" "$OUT/sum/${PFX}_train_kernel_stats.csv" 8 "python3 bench.py --mode train --steps 6 --warmup 2: nuScenes polar-pillar model, bs = 4 sweeps of 30k points, 8 iterations (2 warm-up + 6); NOTE: weight gradients run on a second stream beside the data-gradient chain -- the durations of overlapping kernels add up to more than the wall time of an iteration"
python3 tools/kernel_stats_summary.py "$OUT/partner" "$OUT/sum/${PFX}_train_partner_kernel_stats.csv" 4 "python3 tools/train_partner_profile.py --steps 3: Waymo PARTNER detector, bs = 2 sweeps of 180k points, 4 iterations (1 warm-up + 3); NOTE: weight gradients run on a second stream -- the durations of overlapping kernels add up to more than the wall time of an iteration" | head -30
[ -n "$KEEP_RAW" ] || rm -rf "$OUT/train" "$OUT/partner"
