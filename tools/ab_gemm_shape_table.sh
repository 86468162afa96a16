#!/bin/bash
# per-shape policy A/B for the stage-1 / stage-2 bias products (T256 and B256)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/ab_gemm_shape_table; mkdir -p $O
X="--steps 10 --warmup 3 --no-companions --no-cpu-baseline --no-fp32-companion --no-graph-companion --no-pmc-traffic --kernel-table"
B="--workload T256 $X"
python bench.py $B > $O/t256_base.json 2> $O/t256_base.err
python tools/policy_ab.py "OWN_SHAPE_TABLE={(192,768):True,(192,576):True,(576,192):True,(768,192):True}" -- $B > $O/t256_s1own.json 2> $O/t256_s1own.err
python tools/policy_ab.py "OWN_SHAPE_TABLE={(384,1536):True,(384,1152):True,(1152,384):True,(384,384):True}" -- $B > $O/t256_s2own.json 2> $O/t256_s2own.err
python bench.py $B > $O/t256_base2.json 2> $O/t256_base2.err
B="--workload B256 $X"
python bench.py $B > $O/b256_base.json 2> $O/b256_base.err
python tools/policy_ab.py "OWN_SHAPE_TABLE={(768,256):True,(256,768):True,(256,1024):True}" -- $B > $O/b256_s1own.json 2> $O/b256_s1own.err
python bench.py $B > $O/b256_base2.json 2> $O/b256_base2.err
for f in t256_base t256_s1own t256_s2own t256_base2 b256_base b256_s1own b256_base2; do python - <<PY
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
    print("$f", d["ms_per_step"], d["value"])
except Exception as e:
    print("$f", "failed", e)
PY
done > $O/summary.txt
cat $O/summary.txt
