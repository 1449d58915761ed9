#!/bin/bash
# usage (GPU box, repo root): tools/mb4_profile.sh <tag> [bench flags]   -- kernel-trace stats of the YAML regime (4 sentences per
# optimizer step), per kernel and per (kernel, workgroups) -> gpurun_out/<tag>/
tag=${1:-mb4}; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats -d $out/kt -o cur -- python bench.py --micro-batch 4 --accum 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-roofline "$@" > $out/bench_profiled.log 2>&1
db=$(find $out/kt -name '*results.db' | head -1)
python tools/rocpd_stats.py $db $out/${tag}_kernel_stats.md > /dev/null 2>&1
python tools/rocpd_stats.py $db $out/${tag}_kernel_stats_by_grid.md --by-grid > /dev/null 2>&1
python - <<PY
import sqlite3
db = sqlite3.connect("$db"); c = db.cursor()
print([r[1] for r in c.execute("pragma table_info(kernels)")])
PY
head -30 $out/${tag}_kernel_stats_by_grid.md
grep '^{' $out/bench_profiled.log | cut -c1-200
rm -rf $out/kt
