#!/bin/bash
# Sample the shader clock / power while bench.py's timed loop runs (diagnostic for DESIGN.md section 5: what the MFMA peak is at
# the clock the chip actually sustains under this workload).  Usage (GPU box): bash tools/clock_under_load.sh > gpurun_out/clock.txt
export PYTHONPATH=kb-ner_amd
python bench.py --steps 160 --warmup 5 --no-cpu-baseline --no-extras > /tmp/bench_clock.json 2>/dev/null &
BP=$!
sleep 9   # import + arena set-up (samples taken before the timed loop show the idle clock)
for i in $(seq 1 30); do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  echo "--- sample $i"
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | head -8
  sleep 0.5
done
wait $BP
tail -1 /tmp/bench_clock.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
echo "--- idle"
sleep 2
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|power" | head -6
