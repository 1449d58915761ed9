"""ModelFinetuner on the MI355X engine: the fine-tuning host loop of KB-NER with data parallelism added.

Behavioural reference (restated): flair/trainers/finetune_trainer.py -- constructor keywords (:51-78), train() keywords
(:379-437) and loop (:780-1348): ColumnDataLoader over train(+dev), batch-order reshuffle per epoch (:819), per micro-batch
loss / gradient_accumulation_steps (:939-946) + backward (:957), every `accum` micro-batches clip_grad_norm_(5.0) (:1010) +
AdamW.step (:1018) + zero_grad + scheduler.step (:1023) with the two lr groups of :552-571 (transitions at lr * lr_rate) and
the linear decay of :686-688 (t_total = ceil(len(loader)/accum) * max_epochs, :679), per-epoch dev evaluation, best / final
model files, optional `save_finetuned_embedding` HF directory (:1289-1312), final_test (:2136).
Clip + AdamW + zero_grad are ONE fused HIP pass (kbner.engine.FusedAdamW); gradients are all-reduced once per step over
RCCL when launched with torchrun (kbner.dp)."""
import logging
import random
import time
from pathlib import Path
from typing import List, Union

import torch
from torch.utils.data.dataset import ConcatDataset

import flair
import flair.nn
from flair.custom_data_loader import ColumnDataLoader
from flair.training_utils import add_file_handler, init_output_file, log_line, store_embeddings

log = logging.getLogger("flair")


def _warn_unknown(where, kwargs):
    """a misspelt YAML key must not vanish into **kwargs (the reference swallows it silently, finetune_trainer.py:436)"""
    known_inert = {"compress_embedding_grad"}
    extra = sorted(k for k in kwargs if k not in known_inert)
    if extra:
        log.warning("%s: ignoring unknown keyword(s) %s -- check the YAML for a misspelt key", where, ", ".join(extra))


class ModelFinetuner:
    def __init__(self, model: flair.nn.Model, teachers: List[flair.nn.Model], corpus, optimizer=None, professors=None,
                 epoch: int = 0, optimizer_state: dict = None, scheduler_state: dict = None, use_tensorboard: bool = False,
                 distill_mode: bool = False, ensemble_distill_mode: bool = False, config=None, train_with_professor: bool = False,
                 is_test: bool = False, language_resample: bool = False, direct_upsample_rate: int = -1, down_sample_amount: int = -1,
                 sentence_level_batch: bool = False, clip_sentences: int = -1, remove_sentences: bool = False,
                 assign_doc_id: bool = False, train_with_doc: bool = False, pretrained_file_dict: dict = None,
                 sentence_level_pretrained_data: bool = False, assign_doc_for_ext_context: bool = False, **kwargs):
        if ensemble_distill_mode or train_with_professor:
            raise NotImplementedError("ensemble_distill_mode / train_with_professor are outside the hot path")
        _warn_unknown("ModelFinetuner", kwargs)
        self.model = model
        # teacher-student knowledge distillation (finetune_trainer.py:223-237): the teachers only label the training set once,
        # before the first epoch (assign_pretrained_teacher_targets), and are dropped after that
        self.distill_mode = bool(distill_mode) and not is_test
        if self.distill_mode:
            if not teachers:
                raise ValueError("distill_mode: true needs at least one teacher (ConfigParser.create_teachers / create_teachers_list)")
            if not (getattr(model, "distill_crf", False) or getattr(model, "distill_exact", False) or
                    getattr(model, "distill_emission", False) or
                    (getattr(model, "distill_posterior", False) and not getattr(model, "multi_view_training", False))):
                # (with none of them the reference's loss for a CRF student is interpolation * 0 + (1 - interpolation) * NLL, :2311,
                # after labelling the whole training set for nothing)
                raise NotImplementedError("distill_mode with a CRF student needs one of the model switches distill_crf / "
                                          "distill_posterior / distill_exact / distill_emission")
            for teacher in teachers:
                teacher.eval()
        self.optimizer_state = optimizer_state     # checkpoint resume (finetune_trainer.py:573,690)
        self.scheduler_state = scheduler_state
        self.corpus = corpus
        self.config = config
        self.teachers = teachers or []
        self.professors = professors or []
        self.epoch = epoch
        self.sentence_level_batch = sentence_level_batch
        self.use_bert = False          # 'bert' is not a substring of 'TransformerWordEmbeddings' 
        self.bert_tokenizer = None
        self.embeddings_storage_mode = "none"
        self.corpus2id = {name: i for i, name in enumerate(getattr(corpus, "targets", []))}
        if direct_upsample_rate > 0 and not is_test and hasattr(corpus, "train_list"):
            # finetune_trainer.py:185-198: replicate each training set `rate` times
            corpus.train_list = [ConcatDataset([d] * int(direct_upsample_rate)) for d in corpus.train_list]
            corpus._train = ConcatDataset(list(corpus.train_list))
        for i, name in enumerate(getattr(corpus, "targets", [])):
            for part in ("train_list", "dev_list", "test_list"):
                for s in getattr(corpus, part)[i]:
                    s.lang_id = i
        self.sentence_level_pretrained_data = sentence_level_pretrained_data
        if assign_doc_id:   # finetune_trainer.py:108-150: group sentences into documents at -DOCSTART- lines
            for i, name in enumerate(getattr(corpus, "targets", [])):
                docs = {}
                for part, lst in (("train_", corpus.train_list), ("dev_", corpus.dev_list), ("test_", corpus.test_list)):
                    self.assign_documents(lst[i], part, docs, name, train_with_doc)
        if getattr(self.model, "multi_view_training", False):
            self._pair_multi_view_corpora()
        if assign_doc_for_ext_context:
            self.assign_ext_context_doc(self.corpus)   # finetune_trainer.py:373-377

    def _pair_multi_view_corpora(self):
        """finetune_trainer.py:316-344: with `multi_view_training`, every corpus whose NAME contains 'doc' is the context view of
        the (one) corpus whose name does not: sentence k of its train / dev / test split gets `orig_sent` = sentence k of the
        source corpus' split.  ('unlabel' corpora -- the semi-supervised variant, :345-365 -- are not implemented.)"""
        source, targets = None, []
        for name in self.corpus2id:
            low = name.lower()
            if "doc" in low:
                targets.append(name)
            elif "unlabel" in low:
                raise NotImplementedError("multi-view training with an unlabeled corpus (%s) is not implemented" % name)
            else:
                source = name
        if source is None or not targets:
            log.warning("multi_view_training: no (source, *doc*) corpus pair among %s -- training without the second view",
                        list(self.corpus2id))
            return
        si = self.corpus2id[source]
        for tname in targets:
            ti = self.corpus2id[tname]
            log.info("%s -> %s", source, tname)
            for part in ("train_list", "dev_list", "test_list"):
                src, dst = getattr(self.corpus, part)[si], getattr(self.corpus, part)[ti]
                if len(src) < len(dst):
                    raise ValueError("multi-view pairing: %s has %d sentences but its source %s only %d (%s)" %
                                     (tname, len(dst), source, len(src), part))
                for k, sentence in enumerate(dst):
                    sentence.orig_sent = src[k]

    @classmethod
    def load_from_checkpoint(cls, checkpoint: dict, corpus, **kwargs):
        """resume from Model.load_checkpoint()'s dict (reference: ModelTrainer.load_from_checkpoint, trainers/trainer.py:582):
        the trainer restarts at checkpoint['epoch'] with the saved Adam moments, step count, LR-schedule position and RNG streams"""
        return cls(checkpoint["model"], None, corpus, epoch=checkpoint["epoch"], optimizer_state=checkpoint["optimizer_state_dict"],
                   scheduler_state=checkpoint["scheduler_state_dict"], **kwargs)

    def _group_weights(self, group, multi_view_rate):
        """per-sentence loss weights of one accumulation group encoded as a single batch: 1 / (|group| * |micro-batch|), i.e.
        the mean over each micro-batch and the 1/accumulate of the group (finetune_trainer.py:939-946).  With multi-view
        training (finetune_trainer.py:909-914,959-966) a micro-batch that carries a second view has its NLL scaled by
        (1 - rate) and adds rate * KL, the KL averaged over ITS paired sentences (:2394-2395) ->
        (weights, (indices into the concatenated group, KL weights) or None)"""
        G = len(group)
        wts, idx, kw, base = [], [], [], 0
        for bt in group:
            # the reference scales the NLL whenever check_multi_view returns the tag tensor (finetune_trainer.py:909-914) -- some
            # sentence has an orig_sent AND an S-X tag occurs anywhere in the batch -- even if no single sentence has both
            mv_batch, sel = self.model.multi_view_plan(bt, with_flag=True) if multi_view_rate is not None else (False, [])
            f = (1.0 - multi_view_rate) if mv_batch else 1.0
            wts += [f / (G * len(bt))] * len(bt)
            if sel:
                idx += [base + k for k in sel]
                kw += [multi_view_rate / (G * len(sel))] * len(sel)
            base += len(bt)
        return wts, ((idx, kw) if idx else None)

    # ------------------------------------------------------------------ teacher-student knowledge distillation
    @property
    def interpolation(self):               # finetune_trainer.py:1350-1355
        return (self.config or {}).get("interpolation", 0.5)

    @property
    def teacher_annealing(self):           # :1356-1361
        return bool((self.config or {}).get("teacher_annealing", False))

    @property
    def anneal_factor(self):               # :1362-1367
        return (self.config or {}).get("anneal_factor", 2)

    def assign_pretrained_teacher_targets(self, coupled_train_data, teachers, best_k=10, mini_batch_size=32):
        """finetune_trainer.py:1515-1910: run every teacher over the training sets it teaches (`teacher.targets`) and store, per
        sentence and teacher,
          distill_crf        the best_k tag sequences of the n-best Viterbi decoder (+ with crf_attention the softmax of their
                             scores)                                                                   (:1600, :1871-1874)
          distill_posterior  forward_var + backward_var of the TEACHER's CRF over its emissions with the START / STOP / <unk>
                             logits lowered by 1e12                                                    (:1627-1634, :1878)
          distill_exact      the tempered pairwise posteriors + start / end scores                     (:1705-1722, :1878-1885)
        All of it is computed on the device (teacher.forward = the HIP encoder + head; kbner_crf_viterbi_nbest /
        kbner_crf_fb_score / kbner_crf_pair_posterior) and kept on the host, trimmed to the sentence.  Every data-parallel rank
        labels the whole training set: the batches a rank trains on change with every epoch's shuffle.
        Returns the flat list of training sentences."""
        import numpy as np
        log.info("Distilling sentences as targets...")
        if len(self.corpus.targets) != len(coupled_train_data):
            raise ValueError("Coupled train data is not equal to target!")
        m = self.model
        counter = 0
        for teacher in teachers:
            if m.tag_dictionary.item2idx != teacher.tag_dictionary.item2idx:
                raise ValueError("the tag_dictionaries of the teacher and student are not same")
            teacher.eval()
            for index, train_data in enumerate(coupled_train_data):
                if self.corpus.targets[index] not in getattr(teacher, "targets", set(self.corpus.targets)):
                    continue
                loader = ColumnDataLoader(list(train_data), mini_batch_size, False, model=teacher,
                                          sentence_level_batch=self.sentence_level_batch)
                loader.assign_tags(teacher.tag_type, teacher.tag_dictionary)
                for batch in loader:
                    counter += len(batch)
                    logits = teacher.forward(batch).contiguous()                       # f32 [B, n, T], device
                    n = logits.shape[1]
                    lens = torch.tensor([len(sn) for sn in batch], dtype=torch.int32, device=logits.device)
                    mask = torch.arange(n, device=logits.device)[None, :] < lens[:, None]
                    if m.distill_crf:
                        path_score, decode_idx = teacher._viterbi_decode_nbest(logits, mask, best_k)
                        decode = (decode_idx * mask[:, :, None]).cpu().numpy().astype(np.int32)
                        path_score = path_score.cpu().numpy()
                    if m.distill_posterior:
                        fb_score = teacher.forward_backward_score(logits, lens).cpu().numpy()
                    if m.distill_exact:
                        pair, s_sc, e_sc = (x.cpu().numpy() for x in teacher.pair_posterior(logits, lens, m.temperature))
                    for i, sentence in enumerate(batch):
                        L = len(sentence)
                        if m.distill_crf:
                            if m.crf_attention:
                                sentence.set_teacher_weights(path_score[i])
                            sentence.set_teacher_target(decode[i, :L])
                        if m.distill_posterior:
                            sentence.set_teacher_posteriors(fb_score[i, :L].copy())
                        if m.distill_exact:
                            sentence.set_teacher_posteriors(pair[i, :max(L - 1, 0)].copy())
                            sentence.set_teacher_startscores(s_sc[i].copy())
                            sentence.set_teacher_endscores(e_sc[i].copy())
                    store_embeddings(batch, "none")
        log.info("Distilled %d sentences", counter)
        return [sn for data in coupled_train_data for sn in data]

    def assign_pretrained_teacher_predictions(self, coupled_train_data, teachers, is_professor=False, faster=False,
                                              mini_batch_size=32):
        """finetune_trainer.py:1417-1513 for sequence labelling: every teacher's emission scores over the training sets it teaches,
        `softmax` of them when the student sets distill_prob (:1474), stored per sentence trimmed to its length (:1492; the
        student's batches zero-pad them, `resort` :2008-2012 = FastSequenceTagger._kd_batch).  Computed on the device
        (teacher.forward = the HIP encoder + head), kept on the host.  Returns the flat list of training sentences."""
        if is_professor or faster:
            raise NotImplementedError("professors / the `faster` batch-level storage are outside the hot path")
        log.info("Distilling sentences...")
        if len(self.corpus.targets) != len(coupled_train_data):
            raise ValueError("Coupled train data is not equal to target!")
        m = self.model
        counter = 0
        for teacher in teachers:
            if m.tag_dictionary.item2idx != teacher.tag_dictionary.item2idx:
                raise ValueError("the tag_dictionaries of the teacher and student are not same")
            teacher.eval()
            for index, train_data in enumerate(coupled_train_data):
                if self.corpus.targets[index] not in getattr(teacher, "targets", set(self.corpus.targets)):
                    continue
                loader = ColumnDataLoader(list(train_data), mini_batch_size, False, model=teacher,
                                          sentence_level_batch=self.sentence_level_batch)
                loader.assign_tags(teacher.tag_type, teacher.tag_dictionary)
                for batch in loader:
                    counter += len(batch)
                    logits = teacher.forward(batch)                                    # f32 [B, n, T], device
                    if m.distill_prob:
                        logits = torch.softmax(logits, -1)
                    logits = logits.cpu().numpy()
                    for i, sentence in enumerate(batch):
                        sentence.set_teacher_prediction(logits[i, :len(sentence)].copy())
                    store_embeddings(batch, "none")
        log.info("Distilled %d sentences", counter)
        return [sn for data in coupled_train_data for sn in data]

    # ------------------------------------------------------------------ training
    def train(self, base_path: Union[Path, str], learning_rate: float = 5e-5, mini_batch_size: int = 32,
              eval_mini_batch_size: int = None, max_epochs: int = 100, anneal_factor: float = 0.5, patience: int = 10,
              min_learning_rate: float = 5e-9, train_with_dev: bool = False, dataset_level_macro_avg: bool = True,
              monitor_train: bool = False, monitor_test: bool = False, embeddings_storage_mode: str = "cpu",
              checkpoint: bool = False, save_final_model: bool = True, anneal_with_restarts: bool = False, shuffle: bool = True,
              true_reshuffle: bool = False, param_selection_mode: bool = False, num_workers: int = 4, sampler=None,
              use_amp: bool = False, use_autocast: bool = False, language_attention_warmup_and_fix: bool = False,
              language_attention_warmup: bool = False, language_attention_entropy: bool = False,
              train_language_attention_by_dev: bool = False, calc_teachers_target_loss: bool = False,
              entropy_loss_rate: float = 1, amp_opt_level: str = "O1", professor_interpolation=0.5, best_k=10,
              max_epochs_without_improvement=100, gold_reward=False, warmup_steps: int = 0, warmup_epochs: int = 1,
              use_warmup: bool = False, gradient_accumulation_steps: int = 1, lr_rate: int = 1, decay: float = 0.75,
              decay_steps: int = 5000, use_unlabeled_data: bool = False, sort_data: bool = True, fine_tune_mode: bool = False,
              debug: bool = False, min_freq: int = -1, min_lemma_freq: int = -1, min_pos_freq: int = -1,
              unlabeled_data_for_zeroshot: bool = False, rootschedule: bool = False, freezing: bool = False,
              save_finetuned_embedding: bool = False, multi_view_rate: float = 0.5, one_by_one: bool = False,
              select_model_by_macro: bool = False, log_interval: int = None, fuse_accumulation: bool = True,
              overlap_allreduce: bool = True, **kwargs) -> dict:
        """Keywords are the reference's (finetune_trainer.py:379-437), spelled out so that a misspelt YAML key is reported
        (`**kwargs` only warns).  Those that select paths outside the hot path raise when switched on; the annealing /
        plateau-scheduler keywords have no effect on the fine-tune path in the reference either (:672-688 uses the linear
        schedule whenever fine_tune_mode), so they are accepted and unused.

        fuse_accumulation (not in the reference): the micro-batches of one gradient-accumulation group are encoded as ONE
        batch whose sentences carry the weights 1/(accumulate * |their micro-batch|) -- the same loss and gradient as
        `loss / accumulate` summed over the group (finetune_trainer.py:939-957), but the KB-NER YAMLs' `mini_batch_size: 1,
        gradient_accumulation_steps: 4` then runs 4 sentences per launch instead of 1 (3.6x on one MI355X).  Dropout streams
        differ (WordDropout positions are shared by the fused batch), nothing else does."""
        from kbner import dp
        from kbner.engine import FusedAdamW
        _warn_unknown("ModelFinetuner.train", kwargs)
        for flag, on in (("use_amp", use_amp), ("use_autocast", use_autocast), ("rootschedule", rootschedule),
                         ("freezing", freezing), ("use_unlabeled_data", use_unlabeled_data),
                         ("unlabeled_data_for_zeroshot", unlabeled_data_for_zeroshot), ("gold_reward", gold_reward),
                         ("language_attention_warmup", language_attention_warmup or language_attention_warmup_and_fix),
                         ("sampler", sampler is not None)):
            if on:
                raise NotImplementedError("train(%s=...) selects a path outside the XLM-R + CRF fine-tune hot path" % flag)
        base_path = Path(base_path)
        base_path.mkdir(parents=True, exist_ok=True)
        is_main = dp.rank() == 0
        handler = add_file_handler(log, base_path / "training.log") if is_main else None
        self.embeddings_storage_mode = embeddings_storage_mode
        eval_bs = eval_mini_batch_size or max(mini_batch_size, 32 if self.sentence_level_batch else mini_batch_size)
        accum = max(1, int(gradient_accumulation_steps))
        W = dp.world_size()

        train_sets = list(self.corpus.train_list)
        if train_with_dev:
            train_sets = [ConcatDataset([t, d]) for t, d in zip(self.corpus.train_list, self.corpus.dev_list)]
        train_data = [s for ds in train_sets for s in ds]
        if self.distill_mode:
            # finetune_trainer.py:597-636: every training sentence gets its teacher targets, then the teachers are released
            if self.model.distill_crf or self.model.distill_posterior or self.model.distill_exact:      # :616-619
                train_data = self.assign_pretrained_teacher_targets(train_sets, self.teachers, best_k=best_k,
                                                                    mini_batch_size=mini_batch_size)
            else:
                train_data = self.assign_pretrained_teacher_predictions(train_sets, self.teachers, mini_batch_size=mini_batch_size)
            # release the teachers: their engines' device arenas (fp32 parameters, bf16 shadow -- ~3.4 GB for an XLM-R-large-sized
            # teacher) go back to the allocator once no reference is left (train.py drops its own list after building the trainer)
            for t in self.teachers:
                release = getattr(t, "release_device_memory", None)
                if release is not None:
                    release()
            self.teachers = []
            if torch.cuda.is_available():
                torch.cuda.empty_cache()
        loader = ColumnDataLoader(train_data, mini_batch_size, shuffle, use_bert=self.use_bert, tokenizer=self.bert_tokenizer,
                                  model=self.model, sentence_level_batch=self.sentence_level_batch, sort_data=sort_data)
        loader.assign_tags(self.model.tag_type, self.model.tag_dictionary)
        dev_loaders = []
        if not train_with_dev:
            for ds in self.corpus.dev_list:
                dl = ColumnDataLoader(list(ds), eval_bs, False, sort_data=sort_data, model=self.model,
                                      sentence_level_batch=self.sentence_level_batch)
                dl.assign_tags(self.model.tag_type, self.model.tag_dictionary)
                dev_loaders.append(dl)
        test_loaders = []
        if monitor_test:
            for ds in self.corpus.test_list:
                tl = ColumnDataLoader(list(ds), eval_bs, False, sort_data=sort_data, model=self.model,
                                      sentence_level_batch=self.sentence_level_batch)
                tl.assign_tags(self.model.tag_type, self.model.tag_dictionary)
                test_loaders.append(tl)

        steps_epoch = dp.steps_per_epoch(len(loader), accum, W)
        t_total = steps_epoch * max_epochs
        # finetune_trainer.py:679-688: `use_warmup` derives the warm-up length from warmup_epochs; otherwise warmup_steps is used as given
        warmup = steps_epoch * int(warmup_epochs) if use_warmup else int(warmup_steps)
        arena = self.model.engine.arena
        # every replica starts from rank 0's parameters: the head / transitions are drawn from each process's own torch seed
        # (sequence_tagger_model.py:402-410 uses the global RNG), and nothing else would ever make them equal
        if W > 1:
            dp.broadcast_params_(arena.p)
            arena.refresh_shadow()
        opt = FusedAdamW(arena, lr=learning_rate, lr_rate=float(lr_rate), eps=1e-6, weight_decay=0.0, max_norm=5.0,
                         t_total=t_total, warmup=warmup)
        self.optimizer = opt
        # word-embedding rows that receive no gradient in a step are updated when the encoder next looks them up instead of being
        # streamed through HBM every step (FusedAdamW.lazy_rows: the same updates, bit for bit; state_dict / save materialize)
        # (trainer.lazy_embedding_rows: True = where it pays, a step that visits at most an eighth of the table; "always"; False)
        lazy_sw = getattr(self, "lazy_embedding_rows", True)
        if lazy_sw == "always" and hasattr(opt, "lazy_rows"):
            opt.lazy_rows = True
        elif bool(lazy_sw) and hasattr(opt, "lazy_rows_for"):
            opt.lazy_rows_for(mini_batch_size * accum * W * 512)   # sub-tokens per step at most: 512 per sentence
        rng = random.Random(20220711)  # rank-shared shuffle of the batch order
        order = list(range(len(loader)))  # batch order as a permutation of the loader's (fixed) batches: checkpointable
        if self.optimizer_state is not None:   # resume (finetune_trainer.py:573,690): Adam moments, step count, RNG streams
            opt.load_state_dict(self.optimizer_state)
            sch = self.scheduler_state or {}
            if sch.get("t_total") not in (None, t_total):
                log.warning("resuming with t_total %s (checkpoint was written with %s)", t_total, sch.get("t_total"))
            if "shuffle_rng" in sch:
                rng.setstate(sch["shuffle_rng"])
            if sch.get("order") is not None and len(sch["order"]) == len(order):
                order = list(sch["order"])
            if "dropout_rng" in sch:
                self.model.engine._drop_rng.bit_generator.state = sch["dropout_rng"]
            self.optimizer_state = self.scheduler_state = None
        elif self.epoch:
            log.warning("starting at epoch %d without optimizer state: Adam moments and the LR schedule restart", self.epoch)
        emb_lo = arena.offsets["emb.word"]
        V, Hh = arena.shapes["emb.word"]
        reducer = dp.GradReducer(arena.g, emb_range=(emb_lo, emb_lo + V * Hh), emb_width=Hh, emb_flags=arena.emb_flags,
                                 compress_embedding=bool(kwargs.get("compress_embedding_grad", False)),
                                 finalize=getattr(arena, "finalize_grads", None)) if W > 1 else None
        if W > 1 and overlap_allreduce:
            self.model.engine.dynamic_tiles = True   # GEMM tiles drawn dynamically: robust to CUs taken by the overlapped collectives
        log_line(log)
        log.info('Model: "XLM-R encoder + linear + CRF on kbner HIP engine", tags=%d', len(self.model.tag_dictionary))
        log.info('Parameters: learning_rate "%s", mini_batch_size "%s", accumulate "%s", max_epochs "%s", world_size "%s", '
                 'global batch "%s", t_total "%s", warmup "%s"', learning_rate, mini_batch_size, accum, max_epochs, W,
                 mini_batch_size * accum * W, t_total, warmup)
        log.info('Model training base path: "%s"', base_path)
        log_line(log)
        loss_txt = init_output_file(base_path, "loss.tsv") if is_main else None
        if is_main:
            with open(loss_txt, "a") as f:
                f.write("EPOCH\tTIMESTAMP\tLEARNING_RATE\tTRAIN_LOSS\tDEV_LOSS\tDEV_F1\tDEV_MACRO_F1\n")

        multi_view = bool(getattr(self.model, "multi_view_training", False))
        if multi_view:
            log.info("multi-view training: (1 - %s) * NLL(context view) + %s * T^2 KL(posterior(context view) || posterior(sentence "
                     "alone)), T = %s", multi_view_rate, multi_view_rate, self.model.temperature)
        if self.distill_mode:
            multi_view = False
            log.info("distill_mode: interpolation * KD(%s) + (1 - interpolation) * NLL, interpolation %s, T = %s",
                     "+".join(k for k in ("distill_posterior", "distill_crf", "distill_exact", "distill_emission")
                              if getattr(self.model, k, False)),
                     "annealed from 1 by %s%% per epoch" % self.anneal_factor if self.teacher_annealing else self.interpolation,
                     self.model.temperature)
        dev_score_history, dev_loss_history, train_loss_history = [], [], []
        best_score, bad_epochs = 0.0, 0   # finetune_trainer.py: best_score starts at 0 and a TIE with it still saves (:1280-1289)
        log_every = log_interval or max(1, len(loader) // W // 10)
        try:
            for epoch in range(self.epoch, max_epochs):
                if shuffle:
                    if true_reshuffle:
                        random.seed(20220711 + epoch)
                        loader.true_reshuffle()
                        order = list(range(len(loader)))
                    else:
                        rng.shuffle(order)
                self.model.train()
                mine = [order[i] for i in dp.shard_indices(len(loader), dp.rank(), W)]
                # the trailing partial accumulation group is averaged over ITS size, not over `accum` (finetune_trainer.py:939-946)
                n_full = len(mine) // accum * accum
                rem = len(mine) - n_full
                losses, scaled, seen, micro = [], [], 0, 0   # scaled: loss / average_factor, what the reference accumulates
                t_ep = t_log = time.time()
                # (teacher annealing gives every micro-batch its own interpolation, finetune_trainer.py:885: a fused group would train
                # all of them with the last one's -- the group is then run micro-batch by micro-batch)
                fuse = bool(fuse_accumulation) and accum > 1 and not (self.distill_mode and self.teacher_annealing)
                # (a softmax student without sentence_loss divides by the micro-batch's TOKEN count, :2539: not a per-sentence weight)
                fuse = fuse and (getattr(self.model, "use_crf", True) or bool(getattr(self.model, "sentence_level_loss", False)))
                group = []
                for local_no, bi in enumerate(mine):
                    batch = loader[bi]
                    seen += len(batch)
                    micro += 1
                    last = local_no == len(mine) - 1
                    flush = micro == accum or last
                    div = accum if local_no < n_full else rem   # this micro-batch's average_factor
                    if reducer is not None and micro == 1:
                        # the word ids of the whole accumulation group are known now: agree on the union of touched
                        # embedding rows on the host while the GPU works (kbner.dp.GradReducer)
                        grp = [loader[b2] for b2 in mine[local_no:local_no + div]]
                        touched = [sn for bt in grp for sn in bt]
                        if multi_view:   # the second view looks its own sub-tokens up
                            touched += [sn.orig_sent for sn in touched if hasattr(sn, "orig_sent")]
                        reducer.begin(self.model.touched_word_ids(touched))
                    hook = reducer.bucket_ready if (reducer is not None and flush and overlap_allreduce) else None
                    kd_ip = None
                    if self.distill_mode:
                        # finetune_trainer.py:882-889: fixed interpolation, or annealed per batch from 1 towards 0
                        kd_ip = self.interpolation
                        if self.teacher_annealing:
                            kd_ip = max(0.0, 1.0 - ((epoch * len(mine) + local_no) / max(len(mine), 1) * self.anneal_factor) / 100.0)
                    kdkw = {"distill_interpolation": kd_ip} if self.distill_mode else {}
                    if fuse:
                        group.append(batch)
                        if flush:
                            sents = [sn for bt in group for sn in bt]
                            wts, mv = self._group_weights(group, multi_view_rate if multi_view else None)
                            fused = self.model.forward_backward(sents, loss_scale=1.0, sentence_weights=wts, grad_ready=hook,
                                                                multi_view=mv, **kdkw)
                            # log the mean of the group's micro-batch losses, once per micro-batch, like the unfused loop
                            losses += [fused] * len(group)
                            scaled += [fused / div] * len(group)
                            group = []
                    elif multi_view:
                        wts, mv = self._group_weights([batch], multi_view_rate)
                        losses.append(self.model.forward_backward(batch, loss_scale=1.0 / div, sentence_weights=wts,
                                                                  grad_ready=hook, multi_view=mv))
                        scaled.append(losses[-1] / div)
                    else:
                        losses.append(self.model.forward_backward(batch, loss_scale=1.0 / div, grad_ready=hook, **kdkw))
                        scaled.append(losses[-1] / div)
                    store_embeddings(batch, embeddings_storage_mode)
                    if flush:
                        scale = reducer.finish() if reducer is not None else 1.0
                        opt.step(grad_scale=scale)
                        micro = 0
                    if (local_no + 1) % log_every == 0 and is_main and len(losses) >= 1:
                        cur = float(torch.stack(losses[-log_every:]).mean())
                        dt = time.time() - t_log
                        log.info("epoch %d - iter %d/%d - loss %.8f - samples/sec: %.2f (x%d ranks)", epoch + 1, local_no + 1,
                                 len(mine), cur, log_every * mini_batch_size / max(dt, 1e-9), W)
                        t_log = time.time()
                # the reference's epoch loss is the mean of loss / average_factor over the micro-batches (:1003-1004,1070), i.e.
                # 1/accum of the mean micro-batch loss; kept, so train_loss_history / loss.tsv read the same
                loss_sum = float(torch.stack(scaled).sum()) if scaled else 0.0
                tot, cnt = dp.all_reduce_scalars([loss_sum, float(len(scaled))])
                train_loss = tot / max(cnt, 1.0)
                train_loss_history.append(train_loss)
                self.model.trained_epochs = epoch + 1
                self.model.eval()
                log_line(log)
                log.info("EPOCH %d done: loss %.4f - lr %.2e - %.1f sentences/sec (all ranks)", epoch + 1, train_loss,
                         learning_rate * opt.lr_lambda(), W * seen / max(time.time() - t_ep, 1e-9))
                # ---- evaluation: the dev / test batches are shared out over the ranks (replicas are identical) and the span
                # counters summed, so every rank holds the same Result; model selection, files and the
                # checkpoint stay on rank 0
                shard = (dp.rank(), W) if W > 1 else None
                score = macro = None
                if dev_loaders:
                    f1s, dls = [], []
                    for name, dl in zip(getattr(self.corpus, "targets", ["dev"]), dev_loaders):
                        res, dl_loss = self.model.evaluate(dl, embeddings_storage_mode=embeddings_storage_mode, shard=shard)
                        if is_main:
                            log.info("%s DEV : loss %.4f - f1 %.4f - macro %.4f", name, dl_loss, res.main_score, res.macro_score)
                        # dataset-level macro average over the dev sets, in PERCENT (:1108-1126)
                        f1s.append((res.macro_score if select_model_by_macro else res.main_score) * 100)
                        dls.append(dl_loss)
                    score = sum(f1s) / len(f1s)
                    if is_main:
                        log.info("Dataset-Level Macro Average: %.2f\tDataset-Level Macro avg loss: %.2f", score, sum(dls) / len(dls))
                    dev_score_history.append(score)
                    dev_loss_history.append(sum(dls) / len(dls))
                for name, tl in zip(getattr(self.corpus, "targets", ["test"]), test_loaders):
                    res, tl_loss = self.model.evaluate(tl, embeddings_storage_mode=embeddings_storage_mode, shard=shard)
                    if is_main:
                        log.info("%s TEST: loss %.4f - f1 %.4f", name, tl_loss, res.main_score)
                if is_main:
                    with open(loss_txt, "a") as f:
                        f.write("%d\t%s\t%.3e\t%.6f\t%s\t%s\t_\n" % (epoch + 1, time.strftime("%H:%M:%S"), learning_rate * opt.lr_lambda(),
                                                                   train_loss, dev_loss_history[-1] if dev_loss_history else "_",
                                                                   score if score is not None else "_"))
                    if score is not None:
                        if score > best_score:
                            best_score, bad_epochs = score, 0
                        else:
                            bad_epochs += 1
                        if score == best_score:   # (:1280-1298) also on a tie with the best so far, e.g. 0.0 in the first epochs
                            log.info("==================Saving the current best model: %s==================", score)
                            self.model.save(base_path / "best-model.pt")
                            if save_finetuned_embedding:
                                self.save_finetuned_embedding(base_path)
                    if checkpoint and not param_selection_mode:
                        # finetune_trainer.py:1261-1277: model + optimizer.state_dict() + scheduler.state_dict() + epoch + loss.
                        # Here: Adam moments + step count, and what the LR schedule / batch order / dropout streams need
                        self.model.save_checkpoint(base_path / "checkpoint.pt", opt.state_dict(),
                                                   {"t": opt.t, "t_total": t_total, "warmup": warmup, "shuffle_rng": rng.getstate(),
                                                    "order": list(order),
                                                    "dropout_rng": self.model.engine._drop_rng.bit_generator.state},
                                                   epoch + 1, train_loss)
                stop = dp.broadcast_object(bad_epochs >= max_epochs_without_improvement if is_main else None)
                if stop:
                    log.info("no improvement for %d epochs: stopping", bad_epochs)
                    break
        except KeyboardInterrupt:
            log_line(log)
            log.info("Exiting from training early.")
        if is_main and save_final_model:
            self.model.save(base_path / "final-model.pt")
            if save_finetuned_embedding and (train_with_dev or not (base_path / "best-model.pt").exists()):
                self.save_finetuned_embedding(base_path)
        dp.barrier()
        final_score = 0.0
        if self.corpus.test is not None and len(self.corpus.test) > 0 and is_main:
            final_score = self.final_test(base_path, eval_bs, quiet_mode=False)
        if handler is not None:
            log.removeHandler(handler)
        return {"test_score": final_score, "dev_score_history": dev_score_history, "train_loss_history": train_loss_history,
                "dev_loss_history": dev_loss_history}

    def save_finetuned_embedding(self, base_path):
        """write `<base_path>/<basename of the embedding dir>/` with tokenizer + current encoder weights, the directory the
        next fine-tuning stage's YAML names (finetune_trainer.py:1289-1312)"""
        for emb in self.model.embeddings.embeddings:
            if getattr(emb, "fine_tune", False) and hasattr(emb, "tokenizer"):
                out = Path(base_path) / Path(str(emb.name)).name
                out.mkdir(parents=True, exist_ok=True)
                emb.tokenizer.save_pretrained(str(out))
                emb.model.save_pretrained(str(out))

    # ------------------------------------------------------------------ testing
    def final_test(self, base_path: Path, eval_mini_batch_size: int, num_workers: int = 8, overall_test: bool = True,
                   quiet_mode: bool = False, nocrf: bool = False, predict_posterior: bool = False, debug: bool = False,
                   keep_embedding: int = -1, sort_data: bool = False, eval_train: bool = False, **kwargs):
        base_path = Path(base_path)
        log_line(log)
        self.model.eval()
        for name in ("best-model.pt", "final-model.pt"):
            if (base_path / name).exists() and getattr(self.model, "engine", None) is not None:
                state = torch.load(str(base_path / name), map_location="cpu", weights_only=False)
                self.model.engine.load_hf_state_dict(state["encoder_state_dict"])
                for k in ("linear.weight", "linear.bias", "transitions"):
                    self.model.engine.set_param(k, state[k])
                log.info("Testing using %s ...", name.split("-")[0] + " model")
                break
        scores = []
        parts = list(zip(getattr(self.corpus, "targets", ["test"]), self.corpus.test_list)) if hasattr(self.corpus, "test_list") \
            else [("test", self.corpus.test)]
        for name, ds in parts:
            loader = ColumnDataLoader(list(ds), eval_mini_batch_size, False, sort_data=sort_data, model=self.model,
                                      sentence_level_batch=self.sentence_level_batch)
            loader.assign_tags(self.model.tag_type, self.model.tag_dictionary)
            res, loss = self.model.evaluate(loader, out_path=base_path / ("%s-test.tsv" % name), embeddings_storage_mode="none")
            log.info("%s", name)
            log.info("%s", res.log_line)
            log.info("%s", res.detailed_results)
            scores.append(res.main_score)
        log_line(log)
        return sum(scores) / max(1, len(scores))

    def assign_documents(self, data_list, doc_name, doc_sentence_dict, corpus_name, train_with_doc=False):
        """distillation_trainer.py:655-674: a `-DOCSTART-` sentence opens a new document; with train_with_doc every sentence gets
        `.doc` (the list of its document's sentences), `.doc_pos`, `.doc_name` -- what TransformerWordEmbeddings(v2_doc) reads"""
        doc_idx = -1
        for sentence in data_list:
            if "-DOCSTART-" in sentence[0].text:
                doc_idx += 1
                doc_key = "start"
            else:
                doc_key = corpus_name + doc_name + str(doc_idx)
            if getattr(self, "sentence_level_pretrained_data", False):
                doc_idx += 1
                doc_key = corpus_name + doc_name + str(doc_idx)
            doc_sentence_dict.setdefault(doc_key, []).append(sentence)
            if train_with_doc:
                sentence.doc_name = doc_key
                sentence.doc = doc_sentence_dict[doc_key]
                sentence.doc_pos = len(doc_sentence_dict[doc_key]) - 1
        return doc_sentence_dict

    def assign_ext_context_doc(self, corpus):
        """config 5 (`assign_doc_for_ext_context: true`, distillation_trainer.py:675-686): every sentence keeps an unchunked copy
        as `sentence.doc_sent` -- what embeddings with `use_internal_doc` encode (embeddings.py:3116-3117) -- and is itself cut
        at the first `<EOS>` token, so the tagger head (BiLSTM / CRF) only sees the real tokens."""
        import copy
        from torch.utils.data.dataset import ConcatDataset
        for data_lists in (self.corpus.train_list, self.corpus.dev_list, self.corpus.test_list):
            for data_list in data_lists:
                for sentence in data_list:
                    words = [w.text for w in sentence]
                    sentence.doc_sent = copy.deepcopy(sentence)
                    if "<EOS>" in words:
                        sentence.chunk_sentence(0, words.index("<EOS>"))
                    sentence.doc_pos = 0
        self.corpus._train = ConcatDataset(list(self.corpus.train_list))
        self.corpus._dev = ConcatDataset(list(self.corpus.dev_list))
        self.corpus._test = ConcatDataset(list(self.corpus.test_list))
