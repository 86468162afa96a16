#!/bin/bash
# Builds: heal_swin_amd/lib/libhealswin_exp{1,2,3}.so = the library with csrc/gemm_nt.hip compiled with -DHS_GEMM_EXP={1,2,3}
# (hipcc ... -DHS_GEMM_EXP=N -c csrc/gemm_nt.hip, linked with the other objects of heal_swin_amd/build/).
cd ${GRAFT_REPO_ROOT:-$(pwd)}
L=heal_swin_amd/lib
cp $L/libhealswin.so $L/libhealswin_base.so
for r in 1 2; do
  for v in base exp1 exp2 exp3; do
    cp $L/libhealswin_$v.so $L/libhealswin.so
    case $v in base) t="shipped kernel";; exp1) t="DMA issued by 4 of 8 waves";; exp2) t="stores under the k-steps";; exp3) t="both";; esac
    python tools/gemm_overlap_premise.py "$t" 2>&1 | grep -v amdgpu.ids
  done
done
cp $L/libhealswin_base.so $L/libhealswin.so
