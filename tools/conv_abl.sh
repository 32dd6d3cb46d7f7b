#!/bin/bash
# experiment sweep of the conv main loop (tuning only; outputs are wrong for abl bits 1..8)
for abl in ${ABLS:-0 16 32 48}; do
  echo "== PN_CONV_ABL=$abl"
  PN_CONV_ABL=$abl PN_CONV_TILE=${TILE:-1} python tools/conv_bench.py --tile ${TILE:-1} --iters 20 2>/dev/null | head -${ROWS:-2}
done
