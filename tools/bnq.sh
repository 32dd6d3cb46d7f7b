#!/bin/bash
# Upper bound of what BatchNorm-statistics fusion could save in the training iteration: builds of norm.hip WITHOUT the forward (1) / backward (2)
# statistics pass (-DPN_BN_EXP=bits; wrong results, times only), each timed with bench.py --mode train; the product build is restored at the end.
#   tools/bnq.sh 0 1 2 3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT/partner_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function"
for v in "$@"; do
  /opt/rocm/bin/hipcc $FL -DPN_BN_EXP=$v -c norm.hip -o ../lib/norm.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libpartner_hip.so ../lib/*.o -ldl
  cd $ROOT
  for rep in 1 2; do
    python3 bench.py --mode train --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PN_BN_EXP=$v', d.get('ms_per_step'), d.get('unit'))"
  done
  cd partner_amd/csrc
done
/opt/rocm/bin/hipcc $FL -c norm.hip -o ../lib/norm.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libpartner_hip.so ../lib/*.o -ldl
