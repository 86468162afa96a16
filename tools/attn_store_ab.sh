#!/bin/bash
# Same-box A/B of the two store-placement changes in the attention kernels (round 3): builds libhealswin.so four times on the
# GPU box and runs tools/bench_attn.py on each.  usage: bash tools/attn_store_ab.sh > gpurun_out/r03_attn_store_ab.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "1 1" "0 1" "1 0" "0 0" "1 1"; do
  set -- $cfg
  echo "=== forward: rows claimed before the stores = $1, backward: deferred stores = $2"
  HS_EXTRA_CXXFLAGS="-DHS_ATTN_FWD_CLAIM=$1 -DHS_ATTN_BWD_DEFER=$2" python heal_swin_amd/build.py --force > /dev/null 2>&1
  python tools/bench_attn.py 2>/dev/null | grep stage
done
python heal_swin_amd/build.py --force > /dev/null 2>&1
