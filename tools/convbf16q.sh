#!/bin/bash
# conv_bf16.hip alone (tools/micro/conv_bf16_check.hip): results vs a naive kernel + launch times on the Waymo layers.
#   TILES="-1 0 2" EXPS="0 1 2" bash tools/convbf16q.sh [reps] [first cases only]   (-1 = the library's own tile choice; EXPS: PN_CB_EXP ablations)
cd "$(dirname "$0")/.."
for e in ${EXPS:-0}; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude -DPN_CB_EXP=$e ${DEFS} tools/micro/conv_bf16_check.hip partner_amd/csrc/pn_common.hip -o /tmp/conv_bf16_check_$e 2>&1 | grep -E "error" | head
  for t in ${TILES:--1}; do
    echo "== exp $e tile $t"
    if [ "$t" = "-1" ]; then /tmp/conv_bf16_check_$e ${1:-20} ${2:-99}; else PN_CONV_BF16_TILE=$t /tmp/conv_bf16_check_$e ${1:-20} ${2:-99}; fi
  done
done
