#!/bin/bash
# rocprofv3 counter passes over the SHIPPED attention kernels (one counter group per run, --kernel-trace only beside --pmc):
#   1. FETCH_SIZE / WRITE_SIZE over tools/bench_attn.py --iters 2  -> gpurun_out/${R}_attn_pmc_hbm_traffic.json
#   (R=r04 by default: the output names carry the round)
#   2. MFMA-busy / wait / LDS counters over tools/bench_attn_one.py at stages 0 and 2 -> gpurun_out/${R}_attn_pmc_mfma_busy_*.json
# usage (on the GPU box, from the repo root): bash tools/collect_attn_pmc.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=${R:-r04}
OUT=$ROOT/gpurun_out/${R}_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o t -- python $ROOT/tools/bench_attn.py --iters 2 > $OUT/$c.log 2>&1
done
F=$(find $OUT/FETCH_SIZE -name '*counter_collection.csv' | head -1)
W=$(find $OUT/WRITE_SIZE -name '*counter_collection.csv' | head -1)
K=$(find $OUT/FETCH_SIZE -name '*kernel_trace.csv' | head -1)
python $ROOT/tools/attn_pmc_traffic.py $F $W $K > $ROOT/gpurun_out/${R}_attn_pmc_hbm_traffic.json 2> $OUT/traffic.err
tail -3 $OUT/traffic.err
for stage in 0 2; do
  i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU" "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $OUT/busy_s${stage}_g$i -o t -- python $ROOT/tools/bench_attn_one.py $stage 3 > $OUT/busy_s${stage}_g$i.log 2>&1
    python $ROOT/tools/pmc_db.py $OUT/busy_s${stage}_g$i attn_ --json > $ROOT/gpurun_out/${R}_attn_pmc_busy_s${stage}_g$i.json 2>> $OUT/busy.err
  done
done
ls -la $ROOT/gpurun_out/ | grep ${R}_attn_pmc
