#!/bin/bash
# Registers / scratch per kernel of one translation unit: tools/kernel_resources.sh heal_swin_amd/csrc/gemm_nt.hip [filter]
src=$1; filt=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics $HS_EXTRA_CXXFLAGS -x hip -c "$src" -o /tmp/kr_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "Function Name|    VGPRs:|ScratchSize|SGPRs:|LDS Size" \
  | sed -E 's/.*(Name: [^ ]*|VGPRs: [0-9]+|SGPRs: [0-9]+|ScratchSize \[bytes\/lane\]: [0-9]+|LDS Size \[bytes\/block\]: [0-9]+).*/\1/' \
  | awk '/^Name/{if(l)print l; l=$0; next}{l=l"\t"$0}END{print l}' | grep -E "$filt" | while read -r line; do
      n=$(echo "$line" | sed -E 's/Name: ([^\t]*).*/\1/' | c++filt | sed -E 's/hs::\(anonymous namespace\):://; s/\(hs::.*//'); echo -e "$n\t$(echo "$line" | cut -f2-)"; done
rm -f /tmp/kr_$$.o
