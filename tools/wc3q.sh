#!/bin/bash
# ablation builds of conv_wchain3_kernel alone (tools/micro/wchain3_ablate.hip): each argument "<PN_WC3_EXP> <PN_WCHAIN_EXP>" is one variant
#   tools/wc3q.sh "0 0" "1 0" "2 0" "4 0" "6 0" "0 6" "7 7"        (SIZE=128 for the 128 x 128 map)
cd "$(dirname "$0")/micro"
for v in "$@"; do
  set -- $v
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DPN_WC3_EXP=$1 -DPN_WCHAIN_EXP=$2 -I../../include wchain3_ablate.hip -o /tmp/wc3_$1_$2 2>/dev/null && /tmp/wc3_$1_$2 ${SIZE:-256}
done
