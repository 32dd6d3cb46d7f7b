#!/bin/bash
# ablations of the F(4,3) K-split kernel's K step (tools/micro/wino4_stamps.hip; PN_WINO4_EXP bits: 1 no transform, 2 no barrier, 4 no fragment
# reads, 8 no LDS stores, 16 no loads, 32 no weight loads, 64 no input loads):  EXPS="0 32 64 16" bash tools/stampq.sh
cd "$(dirname "$0")/micro"
for e in ${EXPS:-0}; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DPN_WINO4_EXP=$e -I../../include wino4_stamps.hip -o /tmp/wino4_stamps_$e 2>/dev/null && echo "EXP $e" && /tmp/wino4_stamps_$e | head -4
done
