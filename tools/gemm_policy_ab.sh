#!/bin/bash
# Same-box A/B of the own-kernel / library choice per product class (whole B@256 step): GELU-forward epilogue up to K, bias
# products up to K.  HS_POLICY_SET="gelu:bias gelu:bias ..." overrides the list.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for v in ${HS_POLICY_SET:-256:0 512:0 1024:0 256:256 256:512 512:512 1024:512 256:0}; do
  g=${v%%:*}; b=${v##*:}
  echo "=== HS_OWN_GELU_MAX_K=$g HS_OWN_BIAS_MAX_K=$b"
  HS_OWN_GELU_MAX_K=$g HS_OWN_BIAS_MAX_K=$b timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-companion --no-graph-companion 2>/dev/null | cut -c1-200
done
