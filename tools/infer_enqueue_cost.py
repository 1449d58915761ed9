import os, sys, time, json
sys.path.insert(0, "tools"); sys.path.insert(0, "kb-ner_amd"); sys.path.insert(0, "tests")
import torch
import train_throughput as tt
tagger, trainer, cc, td, d, sub = tt.setup(sentences=64)
sents = list(cc.train)[:32]
tagger.eval()
tagger.embeddings.embed(sents)
hb, db = tagger._device_batch(sents)
eng = tagger.engine
for _ in range(4):
    eng.forward_features(db)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    eng.forward_features(db)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(json.dumps({"graph": os.environ.get("KBNER_INFER_GRAPH", "1"), "host_enqueue_ms_per_forward": (t1 - t0) / 10 * 1e3, "device_ms_per_forward": (t2 - t0) / 10 * 1e3}))
