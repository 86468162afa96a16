#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_attn_module.py tests/test_gpu_baseline_configs.py -q 2>&1 | grep -E "^E  .*AssertionError|passed|failed|^FAILED|Error" | cut -c1-220 | head -20
TAG=r05b bash tools/collect_attn_pmc_T256.sh 2>&1 | head -12
