#!/bin/bash
# usage (GPU box, repo root): tools/wgrad_pmc.sh <outdir>   -- L2 / fabric counters of the grouped weight-gradient launch per variant
out=${1:-gpurun_out/r5/wgrad}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*" | sort -u > $out/tcc_counters.txt
i=0
# (WGRAD_VARIANTS, e.g. "0,1,3,7", reaches tools/wgrad_lab.py through the environment: which main loops to run)
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum"; do
  i=$((i+1))
  for tok in ${WGRAD_TOKENS_LIST:-65536 16384}; do
    WGRAD_TOKENS=$tok WGRAD_REPS=1 timeout 200 rocprofv3 --kernel-trace --pmc $set -d $out/p${i}_$tok --output-format csv -- python tools/wgrad_lab.py > $out/p${i}_$tok.log 2>&1 || echo "pass $i ($set) tokens $tok failed"
  done
done
python - <<PY
import collections, csv, glob, os, re
out = "$out"
for d in sorted(glob.glob(os.path.join(out, "p*_*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(float); name = {}
        for r in csv.DictReader(open(f)):
            per[(int(r["Dispatch_Id"]), r["Counter_Name"])] += float(r["Counter_Value"])
            name[int(r["Dispatch_Id"])] = re.sub(r"\(.*", "", r["Kernel_Name"])[:40]
        for (disp, c), v in sorted(per.items()):
            if "gemm256" in name[disp]:
                print(os.path.basename(d), disp, name[disp], c, "%.4g" % v)
PY
