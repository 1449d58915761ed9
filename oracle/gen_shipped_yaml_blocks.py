#!/usr/bin/env python
"""TEST INFRASTRUCTURE (container only).  The blocks of every YAML the reference ships (`/root/reference/config/*.yaml`) that
reach the hot path's constructors -- `model:`, `train:`, the trainer's own block, `embeddings:` and the corpus entries of the
target -- as DATA (keys and values), so that a CPU test can check that the mirror's signatures accept every key a KB-NER user's
config can contain (a misspelt or unsupported key must be reported, never swallowed).  Writes tests/golden/shipped_yaml_blocks.json.
python oracle/gen_shipped_yaml_blocks.py"""
import glob
import json
import os

import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    out = {}
    for f in sorted(glob.glob("/root/reference/config/*.yaml")):
        c = yaml.safe_load(open(f))
        trainer = c.get("trainer", "ModelFinetuner")
        target = c.get("targets", "ner")
        tgt = c.get(target) or {}
        corpora = {k: v for k, v in tgt.items() if k.startswith("ColumnCorpus-")}
        out[os.path.basename(f)] = {
            "trainer": trainer, "trainer_block": c.get(trainer) or {}, "model": c.get("model") or {}, "train": c.get("train") or {},
            "embeddings": c.get("embeddings") or {}, "targets": target, "Corpus": tgt.get("Corpus"), "corpora": corpora,
            "top_level_keys": sorted(c),
        }
    path = os.path.join(GOLD, "shipped_yaml_blocks.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "configs")


if __name__ == "__main__":
    main()
