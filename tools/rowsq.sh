#!/bin/bash
# Ablation builds of pillar_rows_kernel (-DPN_ROWS_EXP=bits, see csrc/pillar_rows.hip), each timed inside the nuScenes frame (one frame in flight):
#   tools/rowsq.sh 0 1 2 4 8
# (on the GPU box; rebuilds pillar_rows.o + the library per variant and restores the product build at the end)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT/partner_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function"
for v in "$@"; do
  /opt/rocm/bin/hipcc $FL -DPN_ROWS_EXP=$v -c pillar_rows.hip -o ../lib/pillar_rows.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libpartner_hip.so ../lib/*.o
  echo "PN_ROWS_EXP=$v"
  cd $ROOT; bash tools/prof_stats.sh rowsq --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-train-leg --no-roofline-events --no-batched --no-c4 --no-c5 --no-sustained 2>&1 | grep "pillar_rows_kernel" | cut -c1-140; cd partner_amd/csrc
done
/opt/rocm/bin/hipcc $FL -c pillar_rows.hip -o ../lib/pillar_rows.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libpartner_hip.so ../lib/*.o
