cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "== variant 4 prof"; KBNER_ATTN_PROF=1 KBNER_ATTN=4 LAB_NO_BWD=1 timeout 300 tools/micro/attn_lab 0 128 512 3
echo "== variant 4 prof nodma"; KBNER_ATTN_DBG=1 KBNER_ATTN_PROF=1 KBNER_ATTN=4 LAB_NO_BWD=1 timeout 300 tools/micro/attn_lab 0 128 512 3
} > gpurun_out/lab5.log 2>&1
cat gpurun_out/lab5.log
