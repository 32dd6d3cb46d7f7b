#!/bin/bash
# Tracked profiles of the PARTNER-specific legs of the bench line:  tools/prof_c4c5.sh r3
#   <pfx>_c4_kernel_stats.csv       Waymo PARTNER cfg, bs = 2, f32, eager (the `c4.f32` object of bench.py)
#   <pfx>_c4_bf16_kernel_stats.csv  the same with bf16 BEV convolutions
#   <pfx>_c4_pmc_mfma.csv           SQ counters of the f32 run (own pass: --pmc with --kernel-trace only); <pfx>_c4_bf16_pmc_mfma.csv: of the bf16 run
#   <pfx>_c5_kernel_stats.csv       300k-point streaming frame, raw sweeps -> boxes
# The program goes directly after `--`.
PFX=${1:-r3}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_c4c5_$PFX
mkdir -p "$OUT/sum"
cd /tmp && export TMPDIR=/tmp
P="$ROOT/tools/c4_e2e_profile.py"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c4_f32" -o s -- python3 $P f32 10 2 > "$OUT/c4_f32.log" 2> "$OUT/c4_f32.err" || tail -5 "$OUT/c4_f32.err"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c4_bf16" -o s -- python3 $P bf16 10 2 > "$OUT/c4_bf16.log" 2> "$OUT/c4_bf16.err" || tail -5 "$OUT/c4_bf16.err"
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$OUT/c4_sq" -o q -- python3 $P f32 3 2 > /dev/null 2> "$OUT/c4_sq.err" || echo "SQ pass failed"
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$OUT/c4_sq16" -o q -- python3 $P bf16 3 2 > /dev/null 2> "$OUT/c4_sq16.err" || echo "SQ pass (bf16) failed"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c5" -o s -- python3 "$ROOT/tools/c5_profile.py" > "$OUT/c5.log" 2> "$OUT/c5.err" || tail -5 "$OUT/c5.err"
cat "$OUT/c4_f32.log" "$OUT/c4_bf16.log"
cd "$ROOT"
python3 tools/kernel_stats_summary.py "$OUT/c4_f32" "$OUT/sum/${PFX}_c4_kernel_stats.csv" 13 "tools/c4_e2e_profile.py f32 10 2: Waymo PARTNER cfg, 2 sweeps of 180k points per step, f32, eager launches (13 steps incl. 3 warm-ups)"
python3 tools/kernel_stats_summary.py "$OUT/c4_bf16" "$OUT/sum/${PFX}_c4_bf16_kernel_stats.csv" 13 "tools/c4_e2e_profile.py bf16 10 2: the same with bf16 BEV convolutions (13 steps incl. 3 warm-ups)" | head -3
python3 tools/kernel_stats_summary.py --mfma "$OUT/c4_sq" "$OUT/sum/${PFX}_c4_pmc_mfma.csv" | head -40
python3 tools/kernel_stats_summary.py --mfma "$OUT/c4_sq16" "$OUT/sum/${PFX}_c4_bf16_pmc_mfma.csv" | head -24
python3 tools/kernel_stats_summary.py "$OUT/c5" "$OUT/sum/${PFX}_c5_kernel_stats.csv" 13 "tools/c5_profile.py: nuScenes 10-sweep frame (300k raw points) -> boxes, eager launches of the StreamingFrameEngine step (13 frames)" | head -30
ls "$OUT/sum"
[ -n "$KEEP_RAW" ] || rm -rf "$OUT/c4_f32" "$OUT/c4_bf16" "$OUT/c4_sq" "$OUT/c4_sq16" "$OUT/c5"   # gpurun copies back at most 64 MiB
