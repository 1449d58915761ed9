cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
timeout 200 python tools/adamw_bench.py --reps 10 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5/smoke2.log 2>&1; echo smoke rc=$?
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r5/mb4 -o cur -- python bench.py --micro-batch 4 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/r5/mb4.log 2>&1
python tools/rocpd_stats.py $(find gpurun_out/r5/mb4 -name '*results.db' | head -1) gpurun_out/r5/round5_mb4x1_kernel_stats.md > /dev/null 2>&1; head -30 gpurun_out/r5/round5_mb4x1_kernel_stats.md; grep '^{' gpurun_out/r5/mb4.log | cut -c1-200; rm -rf gpurun_out/r5/mb4
for i in 1 2; do timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['config']['micro_batch'], d['value'], d['ms_per_step'], d['mfma_fraction_end_to_end'], d['roofline']['frac'], d['roofline']['traffic'])"; done
