#!/bin/bash
# usage (GPU box, repo root): tools/gemm128x_pmc.sh <outdir> [abl ids]  -- L2 / fabric counters of the FFN-up forward GEMM on the
# 256-row ring kernel and on gemm128x (lab library: -DX128_LAB build of csrc/gemm128x.hip, kb-ner_amd/kbner/_exp/libkbner_lab.so)
out=${1:-gpurun_out/r6/x_pmc}
abl=${2:-2,3}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  KBNER_LIB=kb-ner_amd/kbner/_exp/libkbner_lab.so timeout 200 rocprofv3 --kernel-trace --pmc $set -d $out/p$i --output-format csv -- python tools/gemm128x_lab.py --sentences 256 --reps 1 --abl $abl > $out/p$i.log 2>&1 || echo "pass $i ($set) failed"
done
python - <<PY
import collections, csv, glob, os, re
out = "$out"
tab = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(out, "p*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gemm" not in r["Kernel_Name"]: continue
            nm = re.sub(r"^void ", "", r["Kernel_Name"]); nm = re.sub(r"\(.*", "", nm)[:60]
            per[(nm, r["Counter_Name"], int(r["Dispatch_Id"]))].append(float(r["Counter_Value"]))
        agg = collections.defaultdict(list)
        for (nm, c, disp), v in per.items():
            agg[(nm, c)].append(sum(v))
        for (nm, c), v in agg.items():
            tab[nm][c] = sorted(v)[len(v) // 2]     # median over the launches of that kernel
for nm in sorted(tab):
    print(nm)
    for c in sorted(tab[nm]):
        print("    %-28s %.5g" % (c, tab[nm][c]))
PY
