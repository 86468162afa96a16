#!/bin/bash
# Build a VARIANT of libhealswin.so from the current sources with extra compiler flags, without touching the tree's own build:
#   tools/build_variant.sh trace -DHS_GEMM_TRACE        -> heal_swin_amd/build/variant_trace.so
# (ships to the GPU box with the snapshot; tools/ab_two_builds.sh or a plain `cp` puts it in place there)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=/tmp/hs_variant_$name
mkdir -p $tmp/heal_swin_amd $tmp/include
rm -rf $tmp/heal_swin_amd/csrc && cp -r $root/heal_swin_amd/csrc $tmp/heal_swin_amd/
cp $root/heal_swin_amd/build.py $tmp/heal_swin_amd/
cp $root/include/healswin.h $tmp/include/
(cd $tmp && HS_EXTRA_CXXFLAGS="$*" python heal_swin_amd/build.py | tail -1)
cp $tmp/heal_swin_amd/lib/libhealswin.so $root/heal_swin_amd/build/variant_$name.so
echo "built heal_swin_amd/build/variant_$name.so"
