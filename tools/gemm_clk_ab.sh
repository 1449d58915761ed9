#!/bin/bash
# usage: tools/gemm_clk_ab.sh "<flags of lib 1>" "<flags of lib 2>" ...   (CLK_LAYOUTS="0 1 2", CLK_SHAPES as tools/gemm_clk.py)
# builds one -DG2_TRACE library per argument (an empty string = the product code) next to the product library and prints the
# cycles per K step of each, alternating, in ONE gpurun call (same box)
cd /root/repo
mkdir -p kb-ner_amd/kbner/_exp
OBJS=$(ls kb-ner_amd/csrc/build/*.o | grep -v gemm256)
i=0
for flags in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -DG2_TRACE $flags -Wno-inline-asm -c kb-ner_amd/csrc/gemm256.hip -o /tmp/g256_ab$i.o 2>/dev/null || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o kb-ner_amd/kbner/_exp/libkbner_ab$i.so $OBJS /tmp/g256_ab$i.o || exit 1
  i=$((i+1))
done
n=$i
/usr/local/graft/bin/gpurun --timeout 600 -- "for r in 1 2; do for i in \$(seq 0 $((n-1))); do echo LIB \$i; for lay in ${CLK_LAYOUTS:-0 1 2}; do CLK_LAYOUT=\$lay CLK_SHAPES='${CLK_SHAPES:-65536,1024,4096,random;8192,8192,8192,random}' CLK_VARIANTS=${CLK_VARIANTS:-1} KBNER_LIB=\$PWD/kb-ner_amd/kbner/_exp/libkbner_ab\$i.so timeout 120 python tools/gemm_clk.py 2>&1 | grep variant | cut -c1-150; done; done; done" 2>&1 | grep "LIB\|variant"
rm -f kb-ner_amd/kbner/_exp/libkbner_ab*.so
