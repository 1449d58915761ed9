#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_sq.sh <outdir> -- <command...>
# Collects SQ wave-cycle breakdown counters in separate rocprofv3 --pmc passes (8 SQ slots per pass, MI355X_MICROARCH.md
# "rocprofv3 PMC slots"); --kernel-trace only, no other trace domain.
out=$1; shift; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L > $out/counters.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS_VALU_TRANS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $out/p$i --output-format csv -- "$@" > $out/p$i.log 2>&1 || echo "pass $i failed (see $out/p$i.log)"
done
python tools/pmc_sq_summary.py $out
