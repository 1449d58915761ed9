cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export LAB_NO_BWD=1 KBNER_ATTN=4 KBNER_ATTN_ROLL=2
for D in 0 1 6 14 22 30; do echo "== dbg=$D"; KBNER_ATTN_DBG=$D timeout 300 tools/micro/attn_lab 0 128 512 10 | grep "drop=0"; done
