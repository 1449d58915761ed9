#!/bin/bash
# Build ablated variants of the 256^2 GEMM (G2_ABLATE=1..3) next to the product library and time them.
# Usage: tools/gemm_ablate.sh   (builds here, runs on the GPU box through gpurun)
cd /root/repo
mkdir -p kb-ner_amd/kbner/_exp
OBJS=$(ls kb-ner_amd/csrc/build/*.o | grep -v gemm256)
for v in 1 2 3; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -DG2_ABLATE=$v -c kb-ner_amd/csrc/gemm256.hip -o /tmp/g256_$v.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o kb-ner_amd/kbner/_exp/libkbner_ablate$v.so $OBJS /tmp/g256_$v.o || exit 1
done
/usr/local/graft/bin/gpurun --timeout 600 -- 'S="16384,4096,1024;16384,1024,4096;8192,8192,8192"; echo full; python tools/gemm_bench.py --layouts 0,2 --shapes "$S" | grep layout; for v in 1 2 3; do echo ablate $v; KBNER_LIB=$PWD/kb-ner_amd/kbner/_exp/libkbner_ablate$v.so python tools/gemm_bench.py --layouts 0,2 --shapes "$S" | grep layout; done' 2>&1 | tail -30
