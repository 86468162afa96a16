#!/bin/bash
# Evidence for SURVEY 8f N2: time and HBM write bytes of the decoder tail + loss, fused vs unfused (rocprofv3 --pmc WRITE_SIZE, one mode per run).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; R=${R:-r04}; OUT=$ROOT/gpurun_out
python $ROOT/tools/bench_tail.py > $OUT/${R}_tail_fused_vs_unfused.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for m in fused unfused; do
  rm -rf /tmp/tail_$m
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/tail_$m -o t -- python $ROOT/tools/bench_tail.py --mode $m --iters 2 > /tmp/tail_$m.log 2>&1
  python - <<PY >> $OUT/${R}_tail_fused_vs_unfused.txt
import sys; sys.path.insert(0, "$ROOT/tools")
import pmc_db
res = pmc_db.read("/tmp/tail_$m")
tot = 0.0
rows = []
for k, v in res.items():
    if "WRITE_SIZE" in v:
        b = v["WRITE_SIZE"]["avg"] * v["WRITE_SIZE"]["samples"] * 1024 / 4  # KiB per dispatch x dispatches / 4 iterations (2 warm-up + 2)
        rows.append((b, k[:100]))
        tot += b
print(f"\n== $m: HBM bytes WRITTEN per fwd+loss+bwd (rocprofv3 --pmc WRITE_SIZE, KiB units, all kernels of the run / 4 iterations): {tot / 1e6:.0f} MB")
for b, k in sorted(rows, reverse=True)[:8]:
    print(f"   {b / 1e6:9.1f} MB  {k}")
PY
done
cat $OUT/${R}_tail_fused_vs_unfused.txt
