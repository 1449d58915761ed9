#!/usr/bin/env python
"""TEST INFRASTRUCTURE (container only).  Teacher-student knowledge distillation captured by RUNNING THE REFERENCE's own trainer
(imported read-only from /root/reference through oracle/ref_import.py) on the tiny corpus of tests/tiny_assets.py:kd_config --
`ModelFinetuner: {distill_mode: true}`, teachers from `ner.teachers` (`is_teacher_list`), a tiny FROZEN teacher = the tiny
pre-trained encoder + a random head and random CRF transitions, saved as its best-model.pt and loaded back through
ConfigParser.create_teachers_list exactly as train.py does (train.py:98-121).

  kd_e2e.json / .npz   per run (one per KD variant): the teacher's head / transitions; the targets
                       ModelFinetuner.assign_pretrained_teacher_targets (finetune_trainer.py:1515-1910) left on every training
                       sentence (n-best paths, path weights, forward-backward scores; pairwise posteriors + start / end scores
                       for distill_exact); every simple_forward_distillation_loss value (sequence_tagger_model.py:2110-2372) in
                       call order over ModelFinetuner.train (finetune_trainer.py:876-1023, no dropout / shuffling);
                       train_loss_history, dev_score_history; the student's initial and final head / transitions.

The teacher's transitions carry NO -1e12 sentinels: with them the reference's n-best decoder ties on every candidate (see
oracle/gen_golden_kd.py) and its output is not a function of the inputs.
python oracle/gen_golden_kd_e2e.py"""
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

RUNS = {
    "posterior_crf_att": dict(posterior=True, crf=True, attention=True, exact=False, temperature=2.0, interpolation=0.5, best_k=3),
    "exact": dict(posterior=False, crf=False, attention=False, exact=True, temperature=3.0, interpolation=0.7, best_k=2),
}


def head_of(m):
    return {"linear.weight": m.linear.weight.detach().clone().numpy(), "linear.bias": m.linear.bias.detach().clone().numpy(),
            "transitions": m.transitions.detach().clone().numpy()}


def run(name, kw, flair, arrs):
    import tiny_assets
    import yaml
    from flair.config_parser import ConfigParser
    from flair.trainers import ModelFinetuner
    from flair.utils.from_params import Params
    from flair.models.sequence_tagger_model import START_TAG, STOP_TAG
    work = tempfile.mkdtemp(prefix="g15_")
    cfg, tcfg = tiny_assets.kd_config(work, **kw)
    patch_model_dir(os.path.join(work, "xlmr-tiny"))
    with open(os.path.join(work, "cfg.yaml"), "w") as f:
        yaml.safe_dump(cfg, f)
    with open(os.path.join(work, "teacher.yaml"), "w") as f:
        yaml.safe_dump(tcfg, f)
    torch.manual_seed(7)
    cp = ConfigParser(Params.from_file(os.path.join(work, "cfg.yaml")))
    td = cp.tag_dictionary
    T = len(td)
    start, stop = td.get_idx_for_item(START_TAG), td.get_idx_for_item(STOP_TAG)
    # ---- the frozen teacher: built from its YAML, given a head that prefers real tags and sentinel-free transitions, saved
    teacher = cp.create_model(Params.from_file(os.path.join(work, "teacher.yaml")))
    rng = np.random.default_rng(11)
    with torch.no_grad():
        teacher.linear.weight.copy_(torch.from_numpy((rng.standard_normal(tuple(teacher.linear.weight.shape)) * 0.6).astype(np.float32)))
        bias = (rng.standard_normal(T) * 0.5).astype(np.float32)
        bias[[start, stop]] -= 30.0          # a trained teacher never proposes START / STOP as a token's tag
        teacher.linear.bias.copy_(torch.from_numpy(bias))
        teacher.transitions.copy_(torch.from_numpy(rng.standard_normal((T, T)).astype(np.float32)))
    tdir = os.path.join(tcfg["target_dir"], tcfg["model_name"])
    os.makedirs(tdir, exist_ok=True)
    teacher.save(os.path.join(tdir, "best-model.pt"))
    t_head = head_of(teacher)
    del teacher
    teachers = cp.create_teachers_list()
    assert len(teachers) == 1 and np.array_equal(teachers[0].transitions.detach().numpy(), t_head["transitions"])
    torch.manual_seed(1)
    student = cp.create_student()
    init = head_of(student)
    calls = []
    _kd = student.simple_forward_distillation_loss

    def spy(data_points, *a, **k):
        out = _kd(data_points, *a, **k)
        calls.append([float(out), float(k.get("interpolation", -1)), [s.to_tokenized_string() for s in data_points]])
        return out

    student.simple_forward_distillation_loss = spy
    student.save = lambda f: None
    ModelFinetuner.final_test = lambda self, *a, **k: 0.0
    trainer = ModelFinetuner(student, teachers, cp.corpus, config=cp.config, professors=[], **cp.config["ModelFinetuner"])
    out = trainer.train(cp.get_target_path, **cp.config["train"])
    student.simple_forward_distillation_loss = _kd
    # ---- what the teachers left on the training sentences
    sents = []
    for i, s in enumerate(cp.corpus.train_list[0]):
        L = len(s)
        rec = {"text": s.to_tokenized_string(), "len": L}
        pre = "%s/s%d/" % (name, i)
        if kw["crf"]:
            arrs[pre + "target"] = torch.cat(s._teacher_target, -1).cpu().numpy()[:L].astype(np.int32)
            if kw["attention"]:
                arrs[pre + "weights"] = torch.cat(s._teacher_weights, -1).cpu().numpy().astype(np.float32)
        if kw["posterior"]:
            arrs[pre + "fb_score"] = s._teacher_posteriors[0].cpu().numpy()[:L].astype(np.float32)
        if kw["exact"]:
            arrs[pre + "pair"] = s._teacher_posteriors[0].cpu().numpy()[:max(L - 1, 0)].astype(np.float32)
            arrs[pre + "start_score"] = s._teacher_startscores[0].cpu().numpy().astype(np.float32)
            arrs[pre + "end_score"] = s._teacher_endscores[0].cpu().numpy().astype(np.float32)
        sents.append(rec)
    final = head_of(student)
    for k, v in t_head.items():
        arrs["%s/teacher/%s" % (name, k)] = v
    for k, v in init.items():
        arrs["%s/init/%s" % (name, k)] = v
    for k, v in final.items():
        arrs["%s/final/%s" % (name, k)] = v
    shutil.rmtree(work, ignore_errors=True)
    print(name, "calls:", [round(c[0], 4) for c in calls[:8]], "...", len(calls), "train_loss_history", out["train_loss_history"],
          "dev", out["dev_score_history"])
    return {"config_kwargs": kw, "tag_dictionary": td.get_items(), "sentences": sents, "calls": calls,
            "train_loss_history": [float(x) for x in out["train_loss_history"]],
            "dev_score_history": [float(x) for x in out["dev_score_history"]]}


def main():
    flair = ref_import.load_reference()
    ref_import.wrap_auto_tokenizer()
    import transformers
    _orig_am = transformers.AutoModel.from_pretrained
    transformers.AutoModel.from_pretrained = staticmethod(lambda *a, **k: _orig_am(*a, attn_implementation="eager", **k))
    arrs, rec = {}, {}
    for name, kw in RUNS.items():
        rec[name] = run(name, kw, flair, arrs)
    with open(os.path.join(GOLD, "kd_e2e.json"), "w") as f:
        json.dump(rec, f, indent=1, ensure_ascii=False)
    np.savez_compressed(os.path.join(GOLD, "kd_e2e.npz"), **arrs)
    for f in ("kd_e2e.json", "kd_e2e.npz"):
        print("  %-16s %8d bytes" % (f, os.path.getsize(os.path.join(GOLD, f))))


if __name__ == "__main__":
    main()
