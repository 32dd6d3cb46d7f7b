#!/bin/bash
# ablations of conv_wchain2_kernel's K loop (tools/micro/wchain_check.hip; PN_WCHAIN_EXP bits: 1 no height transform, 2 no plane loads, 4 no weight
# loads):  EXPS="0 1 2 4 7" bash tools/wchainq.sh
cd "$(dirname "$0")/micro"
for e in ${EXPS:-0}; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DPN_WCHAIN_EXP=$e -I../../include wchain_check.hip -o /tmp/wchain_check_$e 2>/dev/null && echo "EXP $e" && /tmp/wchain_check_$e 20 | grep -E "^[0-9]|2-D chain -> planes|2-D: 5|stamps|per block|shader clock|last tiles on"
done
