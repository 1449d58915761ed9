#!/usr/bin/env python
"""TEST INFRASTRUCTURE (container only).  Multi-view (cooperative-learning) training captured by RUNNING THE REFERENCE's own
trainer (imported read-only from /root/reference through oracle/ref_import.py) on the tiny paired corpora of
tests/tiny_assets.py:multiview_config -- a plain corpus and its *DOC twin trained jointly with multi_view_training +
distill_posterior + temperature, exactly the shape of the shipped *_doc_joint_multiview_posterior_* YAMLs.

  multiview_e2e.json / .npz   ModelFinetuner.__init__'s corpus pairing (finetune_trainer.py:316-344: which sentence became whose
                              `orig_sent`), then ModelFinetuner.train (finetune_trainer.py:876-1023) for 3 epochs without dropout
                              or shuffling: every forward_loss value (NLL of the batch) and every multi_view_loss value (the
                              posterior KL, :1923,1958-2093) in call order, train_loss_history, dev_score_history, initial and
                              final head / transitions / encoder weights.
python oracle/gen_golden_multiview_e2e.py"""
import json
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden")

from oracle import ref_import  # noqa: E402
from oracle.gen_golden_e2e import patch_model_dir  # noqa: E402

KW = dict(max_epochs=3, accum=2, mini_batch_size=2, temperature=4.0, n_train=12, n_dev=4, n_test=4)


def main():
    flair = ref_import.load_reference()
    ref_import.wrap_auto_tokenizer()
    import tiny_assets
    import yaml
    import transformers
    _orig_am = transformers.AutoModel.from_pretrained
    transformers.AutoModel.from_pretrained = staticmethod(lambda *a, **k: _orig_am(*a, attn_implementation="eager", **k))
    from flair.config_parser import ConfigParser
    from flair.trainers import ModelFinetuner
    from flair.utils.from_params import Params
    import flair.nn

    work = tempfile.mkdtemp(prefix="g14_")
    cfg = tiny_assets.multiview_config(work, **KW)
    patch_model_dir(os.path.join(work, "xlmr-tiny"))
    with open(os.path.join(work, "cfg.yaml"), "w") as f:
        yaml.safe_dump(cfg, f)
    torch.manual_seed(1)
    cp = ConfigParser(Params.from_file(os.path.join(work, "cfg.yaml")))
    student = cp.create_student()
    emb = student.embeddings.embeddings[0]
    init = {"linear.weight": student.linear.weight.detach().clone().numpy(), "linear.bias": student.linear.bias.detach().clone().numpy(),
            "transitions": student.transitions.detach().clone().numpy()}
    init_enc = {k: v.detach().clone().numpy() for k, v in emb.model.state_dict().items()
                if "position_ids" not in k and "token_type_ids" not in k}

    calls = []   # ("nll" | "kl", value, [sentence texts of the batch]) in call order
    _fl, _mv = student.forward_loss, student.multi_view_loss

    def forward_loss(data_points, *a, **k):
        out = _fl(data_points, *a, **k)
        val = out[0] if isinstance(out, tuple) else out
        calls.append(["nll", float(val), [s.to_tokenized_string() for s in data_points]])
        return out

    def multi_view_loss(data_points, *a, **k):
        out = _mv(data_points, *a, **k)
        calls.append(["kl", float(out), [s.to_tokenized_string() for s in data_points]])
        return out

    student.forward_loss, student.multi_view_loss = forward_loss, multi_view_loss
    flair.nn.Model.save = lambda self, f: None
    ModelFinetuner.final_test = lambda self, *a, **k: 0.0
    trainer = ModelFinetuner(student, None, cp.corpus, config=cp.config, **cp.config["ModelFinetuner"])
    pairing = {}
    for name, ci in trainer.corpus2id.items():
        for part in ("train_list", "dev_list", "test_list"):
            pairing["%s/%s" % (name, part)] = [s.orig_sent.to_tokenized_string() if hasattr(s, "orig_sent") else None
                                               for s in getattr(cp.corpus, part)[ci]]
    out = trainer.train(cp.get_target_path, **cp.config["train"])
    student.forward_loss, student.multi_view_loss = _fl, _mv
    final = {"linear.weight": student.linear.weight.detach().numpy(), "linear.bias": student.linear.bias.detach().numpy(),
             "transitions": student.transitions.detach().numpy()}
    enc_sd = {k: v.detach().numpy() for k, v in emb.model.state_dict().items() if "position_ids" not in k and "token_type_ids" not in k}
    rec = {"config_kwargs": KW, "tag_dictionary": cp.tag_dictionary.get_items(), "corpus2id": trainer.corpus2id, "pairing": pairing,
           "calls": calls, "train_loss_history": [float(x) for x in out["train_loss_history"]],
           "dev_score_history": [float(x) for x in out["dev_score_history"]],
           "dev_loss_history": [float(x) for x in out["dev_loss_history"]],
           "multi_view_rate": 0.5, "temperature": float(student.temperature)}
    with open(os.path.join(GOLD, "multiview_e2e.json"), "w") as f:
        json.dump(rec, f, indent=1, ensure_ascii=False)
    arrs = {"init/" + k: v for k, v in init.items()}
    arrs.update({"init_enc/" + k: v for k, v in init_enc.items()})
    arrs.update({"final/" + k: v for k, v in final.items()})
    arrs.update({"final_enc/" + k: v for k, v in enc_sd.items()})
    np.savez_compressed(os.path.join(GOLD, "multiview_e2e.npz"), **arrs)
    shutil.rmtree(work, ignore_errors=True)
    for f in ("multiview_e2e.json", "multiview_e2e.npz"):
        print("  %-24s %8d bytes" % (f, os.path.getsize(os.path.join(GOLD, f))))
    print("calls:", [(c[0], round(c[1], 4)) for c in calls[:12]])
    print("train_loss_history", rec["train_loss_history"], "dev", rec["dev_score_history"])


if __name__ == "__main__":
    main()
