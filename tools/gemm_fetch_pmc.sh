#!/bin/bash
# usage (GPU box, repo root): tools/gemm_fetch_pmc.sh <outdir>  -- L2-miss (fabric-side) read bytes per launch of the engine's K = 1024
# GEMM shapes at 256 sentences on the 256 x 256 ring kernel, against the tile-walk model of DESIGN.md section 3 "GEMM, round 6"
out=${1:-gpurun_out/r6/fetch}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $out/p$i --output-format csv -- python tools/gemm128x_lab.py --sentences 256 --reps 1 --cases ffn_up,qkv,oproj,ffn_down_dgrad > $out/p$i.log 2>&1 || echo "pass $i ($set) failed"
done
python - <<PY
import collections, csv, glob, os, re
out = "$out"
rows = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(out, "p*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(float); meta = {}
        for r in csv.DictReader(open(f)):
            if "gemm256f" not in r["Kernel_Name"]: continue
            per[(int(r["Dispatch_Id"]), r["Counter_Name"])] += float(r["Counter_Value"])
        for (disp, c), v in per.items():
            rows[disp][c] = v
# dispatch order of the lab: per case  [variant 3 check, (variant 19 falls back to the same kernel for non-GELU cases)], then timed launches
disps = sorted(rows)
print("dispatch  FETCH_SIZE x2 [MB]  WRITE_SIZE [MB]  TCC_READ  TCC_MISS  (gemm256f launches in order: ffn_up, qkv, oproj, ffn_down_dgrad blocks)")
for dsp in disps:
    r = rows[dsp]
    print("%6d   %10.1f   %10.1f   %.4g  %.4g" % (dsp, r.get("FETCH_SIZE", 0) * 2 / 1024.0, r.get("WRITE_SIZE", 0) / 1024.0, r.get("TCC_READ_sum", 0), r.get("TCC_MISS_sum", 0)))
PY
