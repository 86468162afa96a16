#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parallel.py tests/test_gpu_hygiene.py -q 2>&1 | tail -5
bash tools/collect_attn_pmc_T256.sh 2>&1 | tail -40
