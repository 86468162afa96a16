#!/bin/bash
# The paper config's attention (HEAL-SWIN-T @ 256 / 8: ring_shift gather tables + cosine attention) next to the same shapes with
# nest_roll + scaled attention (D256): launch times, HBM traffic by PMC, and MFMA-busy / wait / LDS counters at stages 0 and 2.
#   -> gpurun_out/${TAG:-r05}_attn_T256_vs_D256_times.txt, ${TAG:-r05}_attn_pmc_T256_hbm_traffic.json, ${TAG:-r05}_attn_pmc_T256_busy_s{0,2}_g{1..4}.json (+ D256)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_pmcT
mkdir -p $OUT
cd $ROOT
{ for W in T256 D256 T128; do echo "== $W"; python tools/bench_attn.py --workload $W --iters 10 2>/dev/null; done; } > $ROOT/gpurun_out/${TAG:-r05}_attn_T256_vs_D256_times.txt
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o t -- python $ROOT/tools/bench_attn.py --workload T256 --iters 2 > $OUT/$c.log 2>&1
done
F=$(find $OUT/FETCH_SIZE -name '*counter_collection.csv' | head -1)
W=$(find $OUT/WRITE_SIZE -name '*counter_collection.csv' | head -1)
K=$(find $OUT/FETCH_SIZE -name '*kernel_trace.csv' | head -1)
python $ROOT/tools/attn_pmc_traffic.py $F $W $K T256 > $ROOT/gpurun_out/${TAG:-r05}_attn_pmc_T256_hbm_traffic.json 2> $OUT/traffic.err
for WL in T256 D256; do
for stage in 0 2; do
  i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU" "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $OUT/busy_${WL}_s${stage}_g$i -o t -- python $ROOT/tools/bench_attn_one.py $stage 3 bf16 $WL 1 > $OUT/busy_${WL}_s${stage}_g$i.log 2>&1
    python $ROOT/tools/pmc_db.py $OUT/busy_${WL}_s${stage}_g$i attn_ --json > $ROOT/gpurun_out/${TAG:-r05}_attn_pmc_${WL}_busy_s${stage}_g$i.json 2>> $OUT/busy.err
  done
done
done
cat $ROOT/gpurun_out/${TAG:-r05}_attn_T256_vs_D256_times.txt
