#!/bin/bash
# Build the -DG2_TRACE variant of the GEMM next to the product library (git-ignored _exp/) and print the effective shader clock.
cd /root/repo
mkdir -p kb-ner_amd/kbner/_exp
OBJS=$(ls kb-ner_amd/csrc/build/*.o | grep -v gemm256)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -DG2_TRACE -Wno-inline-asm -c kb-ner_amd/csrc/gemm256.hip -o /tmp/g256_trace.o 2>/dev/null || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC -o kb-ner_amd/kbner/_exp/libkbner_trace.so $OBJS /tmp/g256_trace.o || exit 1
/usr/local/graft/bin/gpurun --timeout 600 -- "CLK_LAYOUT=${CLK_LAYOUT:-0} CLK_SHAPES='$CLK_SHAPES' CLK_VARIANTS=$CLK_VARIANTS KBNER_LIB=\$PWD/kb-ner_amd/kbner/_exp/libkbner_trace.so timeout 300 python tools/gemm_clk.py" 2>&1 | grep 'variant'
