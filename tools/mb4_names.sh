#!/bin/bash
# usage (GPU box): tools/mb4_names.sh  -- full names of the torch-native kernels inside the 4-sentence step (rocprofv3 kernel trace)
out=gpurun_out/mb4_names
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace -d $out/kt -o cur -- python bench.py --micro-batch 4 --accum 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-roofline > $out/log.txt 2>&1
python - <<PY
import sqlite3, glob, collections
db = sqlite3.connect(glob.glob("$out/kt/**/*results.db", recursive=True)[0]); c = db.cursor()
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, grid_x, workgroup_x, end-start from kernels").fetchall()
agg = collections.defaultdict(lambda: [0, 0])
for n, g, w, d in rows:
    if "at::" in n or "rocclr" in n:
        k = (n[:400], g // max(w, 1)); agg[k][0] += 1; agg[k][1] += d
for (n, wg), (cnt, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print(cnt, wg, round(tot / cnt / 1e3, 1), "us", n[:330])
PY
rm -rf $out/kt
