#!/bin/bash
# Ablation builds of swv_window_attn_kernel (-DPN_SWV_EXP=bits, see csrc/swin_attn.hip), each timed on the full-size head at bs 1:
#   tools/swvq.sh 0 1 2 4 8 15
# (on the GPU box; rebuilds swin_attn.o + the library per variant and restores the product build at the end)
cd ${GRAFT_REPO_ROOT:-/root/repo}/partner_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function"
for v in "$@"; do
  /opt/rocm/bin/hipcc $FL -DPN_SWV_EXP=$v -c swin_attn.hip -o ../lib/swin_attn.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libpartner_hip.so ../lib/*.o
  echo "PN_SWV_EXP=$v"
  cd ../.. ; bash tools/prof_any.sh swvq 13 tools/c4_head_profile.py head > /dev/null 2>&1; grep swv_window_attn gpurun_out/prof_swvq/sum/swvq_kernel_stats.csv | cut -c1-80; cd partner_amd/csrc
done
/opt/rocm/bin/hipcc $FL -c swin_attn.hip -o ../lib/swin_attn.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libpartner_hip.so ../lib/*.o
