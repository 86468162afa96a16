#!/bin/bash
# rocprofv3 counter passes over hs_gemm_nt's 256x256 tile, role-separated (FAST) vs symmetric DMA issue (HS_GEMM_FAST=0), on the
# stage-2 fc2 (bias) and fc1 (GELU) products: MFMA busy / wave wait / instruction counters, one group per run.
# usage (GPU box, repo root): bash tools/collect_gemm_pmc.sh  -> gpurun_out/r03_gemm_pmc_*.json
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_gemm_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for fast in 1 0; do
  for shape in "98304 512 2048 3 0 fc2_bias" "98304 2048 512 3 1 fc1_gelu"; do
    set -- $shape
    i=0
    for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU" "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
      i=$((i+1))
      d=$OUT/f${fast}_$6_g$i
      HS_GEMM_FAST=$fast timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $d -o t -- python $ROOT/tools/bench_gemm_one.py $1 $2 $3 $4 $5 3 > $d.log 2>&1
      python $ROOT/tools/pmc_db.py $d gemm_nt --json > $ROOT/gpurun_out/r03_gemm_pmc_fast${fast}_$6_g$i.json 2>> $OUT/err.txt
    done
  done
done
ls $ROOT/gpurun_out | grep r03_gemm_pmc_ | head -20
rm -rf $OUT   # the raw rocprofv3 directories are large; the per-group JSON summaries stay
