#!/bin/bash
# usage: tools/run_gpu.sh [bench args...]   -- GPU tests + bench + kernel stats in one gpurun call
cd /root/repo
ARGS="${@:---steps 3 --warmup 1 --no-cpu-baseline}"
/usr/local/graft/bin/gpurun --timeout 1500 -- "mkdir -p gpurun_out/prof; (timeout 600 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -4 | cut -c1-400); cd /tmp && export TMPDIR=/tmp && cd \$GRAFT_REPO_ROOT && (timeout 700 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o cur -- python bench.py $ARGS > gpurun_out/prof/bench.log 2>&1; echo bench rc=\$?); grep '^{' gpurun_out/prof/bench.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"], \"sent/s\", d[\"ms_per_step\"], \"ms/step\", d.get(\"roofline\",{}).get(\"by_layout\"))'" 2>&1 | tail -8
python tools/rocpd_stats.py gpurun_out/prof/cur_results.db 2>/dev/null | head -16 | cut -c1-150
