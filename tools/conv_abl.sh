#!/bin/bash
# ablation sweep of the conv main loop (tuning only; outputs are wrong for abl != 0)
for abl in 0 1 2 3 4 7 8 15; do
  echo "== PN_CONV_ABL=$abl"
  PN_CONV_ABL=$abl PN_CONV_TILE=1 python tools/conv_bench.py --tile 1 --iters 20 | head -2
done
