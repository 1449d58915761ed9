#!/bin/bash
# build + run tools/micro/mfma_power.hip on the GPU box with the shader clock / package power sampled alongside
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_power.hip -o /tmp/mfma_power 2>/dev/null || exit 1
/tmp/mfma_power > /tmp/mfma_power.out &
BP=$!
while kill -0 $BP 2>/dev/null; do
  n=$(wc -l < /tmp/mfma_power.out)
  c=$(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | sed 's/.*(\([0-9]*Mhz\)).*/\1/; s/.*(W): //' | tr '\n' ' ')
  echo "lines_done=$n  $c"
  sleep 0.6
done
cat /tmp/mfma_power.out
