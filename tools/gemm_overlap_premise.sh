#!/bin/bash
# Builds: heal_swin_amd/lib/libhealswin_expN.so = the library with csrc/gemm_nt.hip compiled with -DHS_GEMM_EXP=N
# (hipcc ... -DHS_GEMM_EXP=N -c csrc/gemm_nt.hip, linked with the other objects of heal_swin_amd/build/).  When the experiment ran
# (profiles/archive_r01_r04/r03_gemm_overlap_premise.txt) bit 0 was the role-separated DMA issue, which has since become the shipped kernel (FAST);
# what remains behind the switch is bit 1 (N = 2): the store stream under the k-steps.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
L=heal_swin_amd/lib
cp $L/libhealswin.so $L/libhealswin_base.so
for r in 1 2; do
  for v in base exp2; do
    cp $L/libhealswin_$v.so $L/libhealswin.so
    case $v in base) t="shipped kernel";; exp1) t="DMA issued by 4 of 8 waves";; exp2) t="stores under the k-steps";; exp3) t="both";; esac
    python tools/gemm_overlap_premise.py "$t" 2>&1 | grep -v amdgpu.ids
  done
done
cp $L/libhealswin_base.so $L/libhealswin.so
