#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_mlp_fused.py -q 2>&1 | grep -E "^E  .*AssertionError: (mlp_fused|fused)|passed|failed" | cut -c1-200 > gpurun_out/r05_c4_tests.log 2>&1
cat gpurun_out/r05_c4_tests.log | head -40
