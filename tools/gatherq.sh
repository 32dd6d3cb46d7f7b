#!/bin/bash
# ablations of the gathered MFMA convolution (tools/micro/gather_ablate.hip; PN_GATHER_EXP bits: 1 neighbour = own row, 2 no input loads,
# 4 no weight loads):  EXPS="0 1 2 4" bash tools/gatherq.sh
cd tools/micro
for e in ${EXPS:-0 1 2 4 6}; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DPN_GATHER_EXP=$e -I../../include gather_ablate.hip -o /tmp/ga_$e 2>/dev/null && echo "EXP $e" && /tmp/ga_$e
done
