#!/bin/bash
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
{
echo "=== new tests"; timeout 900 python -m pytest tests/test_gpu_deferred.py tests/test_gpu_optim.py -q 2>&1 | tail -40
echo "=== full gpu suite"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25
} > gpurun_out/r05_c2_tests.log 2>&1
tail -5 gpurun_out/r05_c2_tests.log
