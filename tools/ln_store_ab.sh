#!/bin/bash
# Same-box A/B: non-temporal stores for the residual-stream output of the fused add + LayerNorm kernel (HS_LN_NT_SUM).
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for v in 0 1 0 1; do
  echo "=== HS_LN_NT_SUM=$v"
  HS_EXTRA_CXXFLAGS="-DHS_LN_NT_SUM=$v" python heal_swin_amd/build.py --force > /dev/null 2>&1
  python tools/bench_ln.py 2>/dev/null | grep rows | cut -c1-120
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-companion --no-graph-companion 2>/dev/null | cut -c1-150
done
python heal_swin_amd/build.py --force > /dev/null 2>&1
