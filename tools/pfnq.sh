#!/bin/bash
# Variant builds of pfn.hip (extra compiler flags per variant, e.g. "-DPN_PFN_EXP=1" "-DPN_PFN_NB=128"), each timed alone at N points
# (tools/pfn_bench.py under rocprofv3: per-kernel averages of the tile kernel and the block-per-pillar kernel):
#   N=300000 tools/pfnq.sh "" "-DPN_PFN_EXP=1" "-DPN_PFN_EXP=2"
# (on the GPU box; rebuilds pfn.o + the library per variant and restores the product build at the end)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT/partner_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function"
for v in "$@"; do
  /opt/rocm/bin/hipcc $FL $v -c pfn.hip -o ../lib/pfn.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libpartner_hip.so ../lib/*.o -ldl
  echo "variant [$v]"
  cd $ROOT; bash tools/prof_any.sh pfnq 46 tools/pfn_bench.py 2>&1 | grep -E "^pfn|^index|dynamic_pfn" | cut -c1-150; cd partner_amd/csrc
done
/opt/rocm/bin/hipcc $FL -c pfn.hip -o ../lib/pfn.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libpartner_hip.so ../lib/*.o -ldl
