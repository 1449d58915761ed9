#!/bin/bash
# usage (GPU box, repo root): tools/profile_round.sh <tag>     e.g. round2_a
# kernel-trace stats of the default bench, the unprofiled bench line, and the two HBM-traffic PMC passes -> gpurun_out/<tag>/
tag=${1:-roundX}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 4 --warmup 1 > $out/bench_unprofiled.log 2>&1
grep '^{' $out/bench_unprofiled.log > $out/${tag}_b256x1_bench_unprofiled.json
timeout 400 rocprofv3 --kernel-trace --stats -d $out/kt -o cur -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/bench_profiled.log 2>&1
grep '^{' $out/bench_profiled.log > $out/${tag}_b256x1_bench.json
python tools/rocpd_stats.py $(find $out/kt -name '*results.db' | head -1) $out/${tag}_b256x1_kernel_stats.md > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pf -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $out/pf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pw -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $out/pw.log 2>&1
python tools/pmc_traffic.py $(find $out/pf -name '*counter_collection.csv' | head -1) $(find $out/pw -name '*counter_collection.csv' | head -1) $out/${tag}_hbm_traffic.json | head -8
head -14 $out/${tag}_b256x1_kernel_stats.md
cut -c1-400 $out/${tag}_b256x1_bench_unprofiled.json
rm -rf $out/kt $out/pf $out/pw
