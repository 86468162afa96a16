#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05_call7; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gemm.py tests/test_gpu_hygiene.py -x -q -m gpu > $O/tests.txt 2>&1
tail -2 $O/tests.txt
bash tools/profile_workload.sh r05_f T256 --paper-drop-rates > /dev/null 2>&1
head -45 gpurun_out/r05_f_T256_summary.txt | cut -c1-170
cat gpurun_out/r05_f_T256_bench_under_rocprof.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
X="--workload T256 --steps 10 --warmup 3 --no-companions --no-cpu-baseline --no-fp32-companion --no-graph-companion --no-pmc-traffic"
python bench.py $X 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nodrop', d['ms_per_step'])"
python bench.py $X --paper-drop-rates 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('drop', d['ms_per_step'])"
