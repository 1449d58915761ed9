#!/bin/bash
# usage (GPU box, repo root): tools/ab_step.sh "<variant> <variant> ..."   -- the training step per GEMM main-loop variant, alternating
# processes on one box; per-shape GEMM table of each variant's last run in gpurun_out/r4/shapes_v<variant>.txt
mkdir -p gpurun_out/r4
for v in $1; do
  timeout 300 python bench.py --gemm-variant $v --steps 5 --warmup 2 --no-cpu-baseline --no-extras --gemm-shapes > gpurun_out/r4/ab_$v.json 2> gpurun_out/r4/shapes_v$v.txt
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r4/ab_$v.json") if l.startswith("{")][-1])
print("variant $v: %.1f sentences/s  %.2f ms/step  GEMM family %.1f TFLOP/s  %s" % (d["value"], d["ms_per_step"], d["roofline"]["achieved"], {k: round(x["tflops"]) for k, x in d["roofline"]["by_layout"].items()}))
PY
done
