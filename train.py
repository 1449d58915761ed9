#!/usr/bin/env python
"""train.py -- same command line as the reference's entry script for the XLM-R + CRF path (reference: train.py:35-64 flags,
:81-135 setup, :147-174 --test / --test_speed, :175-410 --parse, :412 train):

    python train.py --config config/<name>.yaml                      # fine-tune
    python train.py --config config/<name>.yaml --test               # evaluate best-model.pt on the test sets
    python train.py --config ... --parse --target_dir D --keep_order # tag CoNLL files under D
    torchrun --nproc-per-node 8 train.py --config ...                # data-parallel over RCCL (new capability)

The reference's own train.py also runs unchanged against this package (put kb-ner_amd/ on PYTHONPATH): it only needs the
flair.* import surface listed in SURVEY.md §8b.  --predict_posterior (marginal decoding), --v2doc (document-window context)
and, for `trainer: ReinforcementTrainer` YAMLs (the ACE stack), --test / --parse with the controller's `best_action` from
training_state.pt are supported; modes outside the hot path (--zeroshot/--all/--other/--predict/--mst/...) are rejected."""
import argparse
import logging
import os
import sys
from pathlib import Path

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))

import flair  # noqa: E402
from flair.config_parser import ConfigParser  # noqa: E402
from flair.custom_data_loader import ColumnDataLoader  # noqa: E402
from flair.datasets import ColumnCorpus  # noqa: E402
from flair.utils.from_params import Params  # noqa: E402

log = logging.getLogger("flair")


def main():
    ap = argparse.ArgumentParser("train.py")
    ap.add_argument("--config", required=True, help="configuration YAML file.")
    ap.add_argument("--test", action="store_true")
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--parse", action="store_true")
    ap.add_argument("--parse_test", action="store_true")
    ap.add_argument("--keep_order", action="store_true")
    ap.add_argument("--target_dir", default="")
    ap.add_argument("--test_speed", action="store_true")
    ap.add_argument("--batch_size", default=-1, type=int)
    ap.add_argument("--num_columns", type=int, default=2)
    ap.add_argument("--comment_symbol", type=str, default=None)
    ap.add_argument("--parse_name", default="")
    ap.add_argument("--output_dir", default="outputs")
    ap.add_argument("--save_embedding", action="store_true")
    for flag in ("zeroshot", "all", "other", "nocrf", "predict", "mst", "predict_posterior", "recur_parse", "v2doc",
                 "parse_train_and_dev", "eval_train", "debug", "remove_x"):
        ap.add_argument("--" + flag, action="store_true")
    args = ap.parse_args()
    for flag in ("zeroshot", "all", "other", "nocrf", "predict", "mst", "recur_parse"):
        if getattr(args, flag):
            sys.exit("--%s is outside the XLM-R + CRF hot path of this build" % flag)
    if args.quiet:
        log.disabled = True

    from kbner import dp
    dp.init_from_env()
    params = Params.from_file(args.config)
    cp = ConfigParser(params, save_embedding=args.save_embedding)
    student = cp.create_student()
    log.info("Model Size: %d", sum(p.numel() for p in student.parameters()))
    corpus = cp.corpus
    trainer_name = cp.config.get("trainer", "ModelFinetuner")
    if trainer_name not in ("ModelFinetuner", "ReinforcementTrainer"):
        sys.exit("trainer %s is outside the hot path" % trainer_name)
    inference = args.test or args.parse or args.parse_test or args.test_speed
    if trainer_name == "ReinforcementTrainer" and not inference:
        sys.exit("ReinforcementTrainer (ACE) YAMLs are supported for --test / --parse only: training the controller is out of scope")
    tcfg = dict(cp.config.get(trainer_name, {}))
    tcfg.setdefault("distill_mode", False)
    if tcfg["distill_mode"] and not inference and trainer_name == "ModelFinetuner":
        # train.py:98-127 of the reference: `is_teacher_list: true` -> <target>.teachers, else one teacher per corpus
        teachers = cp.create_teachers_list() if cp.config.get("is_teacher_list") else cp.create_teachers()
        trainer = flair.trainers.ModelFinetuner(student, teachers, corpus, config=cp.config, professors=[], **tcfg)
        del teachers   # the trainer holds the only reference: it drops the teachers (and their device arenas) once the targets exist
    else:
        if inference:
            tcfg["distill_mode"] = False     # no teachers are built for --test / --parse (train.py:122-123)
        trainer = getattr(flair.trainers, trainer_name)(student, None, corpus, config=cp.config, **tcfg,
                                                        is_test=args.test or args.parse)
    train_config = dict(cp.config["train"])
    base_path = cp.get_target_path
    if args.remove_x:                      # train.py:211-213
        student.remove_x = True
    if args.predict_posterior:             # marginal (forward-backward) decoding instead of Viterbi
        student.predict_posterior = True
    embs = student.embeddings.embeddings if hasattr(student.embeddings, "embeddings") else [student.embeddings]
    if args.v2doc:                         # train.py:223-224: document-window context around every sentence
        for e in embs:
            if hasattr(e, "v2_doc"):
                e.v2_doc = True
    if trainer_name == "ReinforcementTrainer":
        # train.py:214-218: the embedding subset the controller settled on; the stack's own weights come from stack-model.pt
        # ({"rnn": LSTM state dict, "linear.weight", "linear.bias", "transitions"}; INTEGRATION.md shows the two-line export to
        # run next to the reference's checkpoint -- its .pt pickles the reference's classes and cannot be read here)
        import torch
        state = torch.load(str(Path(base_path) / "training_state.pt"), map_location="cpu", weights_only=False)
        student.selection = [int(x) for x in state["best_action"]]
        log.info("Setting embedding mask to the best action: %s (%s)", student.selection, sorted(e.name for e in embs))
        stack_file = Path(base_path) / "stack-model.pt"
        if stack_file.exists():
            student.load_stack_state(torch.load(str(stack_file), map_location="cpu", weights_only=False))
        else:
            log.warning("%s not found: the BiLSTM / CRF head keeps its initial weights", stack_file)
    eval_bs = args.batch_size if args.batch_size > 0 else max(32, int(train_config.get("mini_batch_size", 32)))

    if args.save_embedding:
        trainer.save_finetuned_embedding(base_path)
        return
    if args.test_speed:
        loader = ColumnDataLoader(list(corpus.test), eval_bs, sentence_level_batch=True, model=student)
        loader.assign_tags(student.tag_type, student.tag_dictionary)
        student.evaluate(loader, embeddings_storage_mode="none", speed_test=True)
        return
    if args.test:
        trainer.final_test(base_path, eval_mini_batch_size=eval_bs, quiet_mode=args.quiet, sort_data=not args.keep_order)
        return
    if args.parse or args.parse_test:
        if trainer_name == "ModelFinetuner":
            _load_trained(student, base_path)
        tag_col = student.tag_type
        if args.parse_test:
            sets = list(zip(corpus.targets, corpus.test_list))
        else:
            fmt = {0: "text", 1: "pos", 2: "upos", 3: tag_col} if args.num_columns == 4 else {0: "text", 1: tag_col}
            cc = ColumnCorpus(Path(args.target_dir), fmt, tag_to_bioes=tag_col, comment_symbol=args.comment_symbol)
            sets = [(Path(args.target_dir).name, cc.train)]
        out_dir = Path(args.output_dir)
        out_dir.mkdir(parents=True, exist_ok=True)
        for name, ds in sets:
            loader = ColumnDataLoader(list(ds), eval_bs, sort_data=not args.keep_order, sentence_level_batch=True, model=student)
            loader.assign_tags(student.tag_type, student.tag_dictionary)
            res, _ = student.evaluate(loader, out_path=out_dir / ("%s%s.conllu" % (name, args.parse_name)),
                                      embeddings_storage_mode="none", prediction_mode=True)
            print(name, res.log_line)
        return
    trainer.train(base_path, **train_config)


def _load_trained(student, base_path):
    import torch
    for name in ("best-model.pt", "final-model.pt"):
        f = Path(base_path) / name
        if f.exists():
            st = torch.load(str(f), map_location="cpu", weights_only=False)
            student.engine.load_hf_state_dict(st["encoder_state_dict"])
            for k in ("linear.weight", "linear.bias", "transitions"):
                student.engine.set_param(k, st[k])
            return
    raise FileNotFoundError("no best-model.pt / final-model.pt under %s" % base_path)


if __name__ == "__main__":
    main()
