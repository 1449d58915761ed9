cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export LAB_NO_BWD=1 KBNER_ATTN=4 KBNER_ATTN_ROLL=2
echo "== nodma"; KBNER_ATTN_DBG=1 timeout 300 tools/micro/attn_lab 0 128 512 10
bash tools/pmc_sq.sh gpurun_out/pmc_fwd3 -- tools/micro/attn_lab 0 128 512 3 > gpurun_out/pmc_fwd3.log 2>&1
grep -A26 "attn_fwd3_kernel<false" gpurun_out/pmc_fwd3/summary.txt | head -28
rm -rf gpurun_out/pmc_fwd3/p1 gpurun_out/pmc_fwd3/p2 gpurun_out/pmc_fwd3/p3 gpurun_out/pmc_fwd3/counters.txt
