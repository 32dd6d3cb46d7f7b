#!/bin/bash
# ablation builds of the block-per-group sparse convolution (csrc/sparse_group.hip, PN_SG_EXP bits) timed on the bench frame's own rulebooks:
#   EXPS="0 1 2 8" bash tools/sparseq.sh        (rebuilds sparse_group.o per build; restores the product build at the end)
cd "$(dirname "$0")/.."
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function"
for e in ${EXPS:-0}; do
  rm -f partner_amd/lib/sparse_group.o
  make -C partner_amd/csrc -j8 CXXFLAGS="$FLAGS -DPN_SG_EXP=$e" > /tmp/sg_make.log 2>&1 || tail -5 /tmp/sg_make.log
  echo "== exp $e"
  python tools/sparse_conv_isolated.py 2>&1 | grep -E "${SHOW:- 64-> 64 taps 27| 128->128 taps 27| 64->128|sum }" | head -24
done
rm -f partner_amd/lib/sparse_group.o
make -C partner_amd/csrc -j8 > /tmp/sg_make.log 2>&1
