#!/usr/bin/env python
"""FastSequenceTagger.evaluate end to end with the host / device split per batch (tools/train_throughput.py:evaluate_rate, the
figure bench.py reports as extra.evaluate).  KBNER_INFER_GRAPH=0: eager launches instead of the replayed HIP graph (A/B)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import train_throughput as tt
print(json.dumps(tt.evaluate_rate()))
