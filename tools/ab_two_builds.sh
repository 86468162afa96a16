#!/bin/bash
# Same-box A/B of two builds of libhealswin.so (boxes differ by 3-6 %, so only same-box pairs are comparable):
#   tools/ab_two_builds.sh heal_swin_amd/build/base.so out_prefix 'python tools/bench_gemm_nt.py --variants 3 --only s2'
# runs the command with the tree's library (tag "new") and with the given one (tag "base"), interleaved twice.
# On a gpurun box only: it overwrites heal_swin_amd/lib/libhealswin.so in the (scratch) snapshot.
base=$1; out=$2; shift 2
lib=heal_swin_amd/lib/libhealswin.so
cp $lib /tmp/new_build.so
for rep in 1 2; do
  for tag in new base; do
    if [ $tag = new ]; then cp /tmp/new_build.so $lib; else cp "$base" $lib; fi
    echo "##### $tag (pass $rep)" | tee -a "$out"
    bash -c "$*" 2>&1 | grep -v amdgpu.ids | tee -a "$out"
  done
done
cp /tmp/new_build.so $lib
