#!/bin/bash
# rocprofv3 counter passes (one group per run, --kernel-trace only beside --pmc) over the fused WindowAttention module kernels
# (inference + training form, B stage 0): gpurun_out/${R}_attn_module_pmc_g<i>.json.  usage: R=r04 bash tools/collect_module_pmc.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=${R:-r04}
OUT=$ROOT/gpurun_out/${R}_modpmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU" "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o t -- python $ROOT/tools/bench_attn_module.py --only-fused --case "B stage" --iters 3 > $OUT/g$i.log 2>&1
  python $ROOT/tools/pmc_db.py $OUT/g$i attn_module --json > $ROOT/gpurun_out/${R}_attn_module_pmc_g$i.json 2>> $OUT/err.log
done
ls -la $ROOT/gpurun_out | grep ${R}_attn_module_pmc
