mkdir -p gpurun_out/r5; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout 300 -k "xcd_sync or variants_bit" 2>&1 | tail -3
echo "== wgrad lab (unprofiled)"; WGRAD_VARIANTS=1,3,7 timeout 120 python tools/wgrad_lab.py 2>&1 | tail -7
echo "== wgrad misses"; WGRAD_VARIANTS=1,3,7 WGRAD_REPS=1 timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d gpurun_out/r5/wg4 --output-format csv -- python tools/wgrad_lab.py > gpurun_out/r5/wg4.log 2>&1
python - <<'PY'
import collections, csv, glob, re
for f in glob.glob("gpurun_out/r5/wg4/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float); name = {}
    for r in csv.DictReader(open(f)):
        per[(int(r["Dispatch_Id"]), r["Counter_Name"])] += float(r["Counter_Value"]); name[int(r["Dispatch_Id"])] = re.sub(r"\(.*", "", r["Kernel_Name"])[:44]
    for (d, c), v in sorted(per.items()):
        if "gemm256" in name[d] and c == "TCC_MISS_sum": print(d, name[d], c, "%.4g lines = %.2f GB" % (v, v * 128 / 1e9))
PY
echo "== step A/B gemm variant 1 / 3 / 7"
for v in 1 3 7 1 3 7; do KBNER_GEMM_VARIANT=$v timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gemm variant $v', d['value'], d['ms_per_step'], {k: round(x['tflops']) for k, x in d['roofline']['by_layout'].items()})"; done
