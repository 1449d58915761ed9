"""SequenceTagger / FastSequenceTagger on the MI355X engine: XLM-R encoder -> first-sub-token gather -> linear emissions
-> linear-chain CRF (forward-algorithm loss, Viterbi decode), `use_rnn: false`, `use_crf: true`, `remove_x` honoured.

Behavioural reference (restated): flair/models/sequence_tagger_model.py -- constructor keywords (:100-163), transitions
[to,from] with START row / STOP column = -1e12 (:402-410), forward (:844-1052), forward_loss (:1899-1921),
_calculate_loss incl. remove_x (:2426-2506), _obtain_labels incl. S-X re-padding (:1157-1246), evaluate (:2593-2729:
"token gold pred score" lines, span TP/FP/FN on (tag, str(span)) with the remove_x post-filter :2653-2672), state dict
(:435-477).  All arithmetic runs in libkbner_hip.so through kbner.engine.Tagger; there is no torch autograd graph:
`forward_backward` is the training entry point (forward + explicit backward into the gradient arena)."""
import logging
from pathlib import Path
from typing import List

import os

import numpy as np
import torch

import flair
import flair.nn
from flair.data import Dictionary, Label, Sentence
from flair.training_utils import Metric, Result, store_embeddings

log = logging.getLogger("flair")

START_TAG: str = "<START>"
STOP_TAG: str = "<STOP>"

# constructor switches of the reference that select code outside the hot path (softmax head, MFVI, attention variants, ACE
# controller training): accepted by name, rejected when switched on.  The CRF knowledge-distillation switches (distill_crf,
# crf_attention, distill_with_gold, exp_score, distill_posterior, distill_exact, distill_emission, distill_prob) ARE implemented.
_UNSUPPORTED_TRUE = ("use_mfvi", "use_cnn", "biaf_attention", "use_language_attention",
                     "token_level_attention", "posterior_constraint", "use_language_vector", "enhanced_crf",
                     "use_language_id", "use_transition_attention", "unlabel_entropy_loss", "relearn_embeddings", "map_embeddings",
                     "no_encoder", "new_drop", "use_embedding_masks", "use_gumbel", "embedding_attention",
                     "train_initial_hidden_state")


class SequenceTagger(flair.nn.Model):
    def __init__(self, hidden_size: int, embeddings, tag_dictionary: Dictionary, tag_type: str, use_crf: bool = True,
                 use_mfvi: bool = False, use_rnn: bool = True, use_cnn: bool = False, rnn_layers: int = 1, dropout: float = 0.0,
                 word_dropout: float = 0.05, locked_dropout: float = 0.5, train_initial_hidden_state: bool = False,
                 pickle_module: str = "pickle", interpolation: float = 0.5, sentence_loss: bool = False, distill_crf: bool = False,
                 crf_attention: bool = False, biaf_attention: bool = False, use_language_attention: bool = False,
                 token_level_attention: bool = False, distill_with_gold: bool = False, exp_score: bool = False,
                 distill_posterior: bool = False, distill_prob: bool = False, distill_emission: bool = False,
                 distill_exact: bool = False, posterior_constraint: bool = False, predict_posterior: bool = False,
                 use_language_vector: bool = False, enhanced_crf: bool = False, use_language_id: bool = False,
                 use_transition_attention: bool = False, teacher_hidden: int = 256, num_teachers: int = 0,
                 target_languages: int = 1, gold_const: float = 1.0, posterior_interpolation: float = 0.0, config=None,
                 word_map=None, char_map=None, use_decoder_timer=True, debug=False, unlabel_entropy_loss=False,
                 entropy_loss_rate=0.001, relearn_embeddings=False, map_embeddings=False, no_encoder=False,
                 temperature: float = 1, relearn_size=-1, embedding_selector=False, new_drop: bool = False, use_rl: bool = False,
                 use_embedding_masks: bool = False, use_gumbel: bool = False, embedding_attention: bool = False,
                 testing: bool = False, remove_x: bool = False, multi_view_training: bool = False,
                 calculate_l2_loss: bool = False, l2_loss_only: bool = False, **kwargs):
        """Keywords are the reference's (sequence_tagger_model.py:100-163), spelled out so a misspelt YAML key is reported."""
        super().__init__()
        if kwargs:
            log.warning("SequenceTagger: ignoring unknown keyword(s) %s -- check the YAML for a misspelt key", ", ".join(sorted(kwargs)))
        given = dict(locals())
        for k in _UNSUPPORTED_TRUE:
            if given.get(k):
                raise NotImplementedError("%s=True is outside the MI355X hot path (XLM-R [+ frozen stack + BiLSTM] + linear + CRF)" % k)
        if not use_crf:
            # the softmax student of the reference (sequence_tagger_model.py:2523-2539 loss, :1177-1180 decode), round 6: the fine-tuning
            # tagger only, with none of the losses that are defined on the CRF
            if use_rnn:
                raise NotImplementedError("use_crf=False is implemented for the fine-tuning tagger (use_rnn: false)")
            crf_only = [k for k in ("multi_view_training", "distill_crf", "distill_posterior", "distill_exact", "distill_emission",
                                    "distill_prob", "predict_posterior", "posterior_constraint", "crf_attention") if given.get(k)]
            if crf_only:
                raise NotImplementedError("use_crf=False (softmax head) with %s: these losses / decoders are implemented on the CRF "
                                          "only" % ", ".join(crf_only))
        if use_rnn and rnn_layers != 1:
            raise NotImplementedError("rnn_layers > 1 is not implemented (the KB-NER / ACE configs use the default of 1)")
        # multi-view ("cooperative learning") training: the shipped *_multiview_posterior_* YAMLs set multi_view_training +
        # distill_posterior + temperature; the distill_exact branch and the calculate_l2_loss / l2_loss_only terms of
        # _calculate_multi_view_loss are implemented too (the unlabeled-data branch is not: use_unlabeled_data is refused)
        if multi_view_training and not ((distill_posterior or distill_exact or l2_loss_only) and remove_x and not use_rnn):
            # (with use_crf and none of them the reference's _calculate_multi_view_loss leaves `loss` unbound, :2048-2103)
            raise NotImplementedError("multi_view_training needs distill_posterior (the shipped configs), distill_exact or "
                                      "l2_loss_only, with remove_x: true, use_rnn: false")
        if l2_loss_only and not calculate_l2_loss:
            raise ValueError("l2_loss_only returns the calculate_l2_loss term (sequence_tagger_model.py:2038): enable calculate_l2_loss")
        if (calculate_l2_loss or l2_loss_only) and not multi_view_training:
            raise ValueError("calculate_l2_loss is a term of the multi-view loss: enable multi_view_training")
        # teacher-student knowledge distillation (`distill_mode: true` of ModelFinetuner; simple_forward_distillation_loss below):
        # distill_posterior (without multi_view_training), distill_crf (+ crf_attention, distill_with_gold, exp_score), distill_exact
        # and distill_emission (+ distill_prob): the emission-level KL of :2311-2365, on its own or next to distill_posterior
        kd_student = ((distill_posterior or distill_exact) and not multi_view_training) or distill_crf or distill_emission
        if distill_emission and multi_view_training:
            raise NotImplementedError("distill_emission belongs to distill_mode (teacher-student) training, not to multi_view_training")
        if distill_emission and (distill_crf or distill_exact) and not distill_posterior:
            # the trainer then stores n-best / pairwise targets only (finetune_trainer.py:616-619) and the emission branch stacks
            # empty `_teacher_prediction` lists (sequence_tagger_model.py:2358): the reference fails on this combination
            raise ValueError("distill_emission next to distill_crf / distill_exact needs distill_posterior (whose teacher scores the "
                             "emission term then reads, sequence_tagger_model.py:2349-2353)")
        if distill_emission and distill_posterior and distill_prob:
            # next to distill_posterior the emission term's "teacher prediction" is the teacher's forward-backward SCORES
            # (sequence_tagger_model.py:2349-2353): log-domain, mostly negative, not normalised.  Read as probabilities
            # (distill_prob) the reference's xlogy of a negative target yields NaN; here it would be a finite wrong gradient
            raise ValueError("distill_prob cannot be combined with distill_emission + distill_posterior: the emission term then "
                             "reads the teacher's forward-backward scores, which are not probabilities (the reference's loss is "
                             "NaN for this combination)")
        if kd_student and use_rnn:
            raise NotImplementedError("knowledge distillation is implemented for the fine-tuning student (use_rnn: false)")
        if distill_exact and distill_posterior and not multi_view_training:
            # (multi-view training takes the distill_exact branch when both are set, :2049,2088)   both read / write the sentences' `_teacher_posteriors` (finetune_trainer.py:1880-1886, sequence_tagger_model.py
            # :2128,2163): in the reference the combination fails on the tensor shapes
            raise ValueError("distill_exact and distill_posterior share the teacher-posterior storage: enable one of them")
        if distill_exact and distill_crf:
            # the reference's batch assembly asserts equal target lengths (finetune_trainer.py:1946: the pairwise posteriors have
            # one row less than the n-best paths), so this combination never trains there
            raise ValueError("distill_exact cannot be combined with distill_crf (the reference's resort() asserts on it)")
        if (crf_attention or distill_with_gold) and not distill_crf:
            raise ValueError("crf_attention / distill_with_gold weight the n-best paths of distill_crf: enable distill_crf")
        if distill_with_gold and not crf_attention:
            raise ValueError("distill_with_gold reweights the crf_attention path weights (sequence_tagger_model.py:2282-2304): enable crf_attention")
        self.hidden_size = hidden_size
        self.embeddings = embeddings
        self.tag_dictionary = tag_dictionary
        self.tag_type = tag_type
        self.tagset_size = len(tag_dictionary)
        self.use_crf = bool(use_crf)
        self.use_rnn = bool(use_rnn)
        self.use_cnn = False
        self.sentence_level_loss = sentence_loss
        self.remove_x = remove_x
        self.use_word_dropout = word_dropout  # flair.nn.WordDropout on the token features while training (engine.word_dropout)
        self.use_dropout, self.use_locked_dropout = 0.0, 0.0
        self.predict_posterior = bool(predict_posterior)  # marginal (forward-backward) decoding
        self.temperature = temperature
        self.embedding_selector, self.use_rl = bool(embedding_selector), bool(use_rl)   # set by ReinforcementTrainer (ACE inference)
        self.config = config
        self.target_languages = target_languages
        self.use_decoder_timer = use_decoder_timer
        self.time = 0.0
        self.trained_epochs = 0
        self.use_bert = False
        self.biaf_attention = False
        self.use_language_attention = False
        self.use_language_vector = False
        self.distill_emission, self.distill_prob = bool(distill_emission), bool(distill_prob)
        self.distill_crf, self.distill_exact = bool(distill_crf), bool(distill_exact)
        self.distill_posterior = bool(distill_posterior)
        self.multi_view_training = bool(multi_view_training)
        self.calculate_l2_loss, self.l2_loss_only = bool(calculate_l2_loss), bool(l2_loss_only)
        self.crf_attention, self.distill_with_gold, self.exp_score = bool(crf_attention), bool(distill_with_gold), bool(exp_score)
        self.gold_const = gold_const
        self.selection = None
        self.mask = None
        self.word_map = self.char_map = self.lemma_map = self.postag_map = None
        self.start_idx = tag_dictionary.get_idx_for_item(START_TAG)
        self.stop_idx = tag_dictionary.get_idx_for_item(STOP_TAG)
        self.x_idx = tag_dictionary.get_idx_for_item("S-X") if remove_x else None
        emb = embeddings.embeddings[0] if hasattr(embeddings, "embeddings") else embeddings
        self._emb = emb
        self.engine = None
        self.stack_head = None
        if self.use_rnn:
            # BASELINE config 5: frozen stacked embeddings -> BiLSTM -> linear -> CRF, inference only (kbner.stack)
            if dropout:
                raise NotImplementedError("dropout > 0 on the tagger head is not implemented")
            self._build_stack()
        else:
            if hasattr(embeddings, "embeddings") and len(embeddings.embeddings) != 1:
                raise NotImplementedError("the fine-tuning path (use_rnn: false) takes exactly one TransformerWordEmbeddings; "
                                          "stacked embeddings go with use_rnn: true (inference)")
            if dropout or locked_dropout:
                raise NotImplementedError("dropout / locked_dropout > 0 on the tagger head is not implemented (KB-NER YAMLs use 0.0)")
            self._build_engine()

    # ------------------------------------------------------------------ engine
    def _build_engine(self):
        from kbner import engine as E
        hc = self._emb.model.config
        cfg = E.EncoderConfig(vocab_size=hc.vocab_size, hidden_size=hc.hidden_size, num_hidden_layers=hc.num_hidden_layers,
                              num_attention_heads=hc.num_attention_heads, intermediate_size=hc.intermediate_size,
                              max_position_embeddings=hc.max_position_embeddings, type_vocab_size=getattr(hc, "type_vocab_size", 1),
                              pad_token_id=getattr(hc, "pad_token_id", 1) if getattr(hc, "pad_token_id", 1) is not None else 1,
                              layer_norm_eps=getattr(hc, "layer_norm_eps", 1e-5),
                              hidden_dropout_prob=float(getattr(hc, "hidden_dropout_prob", 0.1)),
                              attention_probs_dropout_prob=float(getattr(hc, "attention_probs_dropout_prob", 0.1)))
        self.engine = E.Tagger(cfg, self.tagset_size, self.start_idx, self.stop_idx, device=flair.device)
        self.engine.use_crf = self.use_crf
        # dropout streams differ per data-parallel rank (each rank sees different sentences anyway)
        self.engine.seed_dropout(int(torch.initial_seed() % (2 ** 31)) + 7919 * int(os.environ.get("RANK", "0")))
        self.engine.word_dropout = float(self.use_word_dropout or 0.0)
        self.engine.load_hf_state_dict(self._emb.model.state_dict())
        g = torch.Generator().manual_seed(int(torch.initial_seed() % (2 ** 31)))
        H = cfg.hidden_size
        bound = 1.0 / (H ** 0.5)
        self.engine.set_param("linear.weight", (torch.rand(self.tagset_size, H, generator=g) * 2 - 1) * bound)
        self.engine.set_param("linear.bias", (torch.rand(self.tagset_size, generator=g) * 2 - 1) * bound)
        tr = torch.randn(self.tagset_size, self.tagset_size, generator=g)
        tr[self.start_idx, :] = -1e12
        tr[:, self.stop_idx] = -1e12
        self.engine.set_param("transitions", tr)
        self._emb.model.source = self.engine
        self._emb.model._state_dict = None  # the arena is the single owner of the weights now

    def drop_gradients(self):
        """forward-only from now on (teachers of a distillation run): the arena's gradient buffer and embedding-row flags are freed"""
        eng = getattr(self, "engine", None)
        if eng is not None and getattr(eng, "arena", None) is not None:
            eng.arena.g = None
            eng.arena.emb_flags = None
            views = eng.arena.__dict__.get("_views")
            if views:      # cached ('g', name) views keep the gradient buffer alive
                eng.arena.__dict__["_views"] = {k: v for k, v in views.items() if k[0] != "g"}

    def release_device_memory(self):
        """drop the engine (parameter arena, bf16 shadow, activation buffers): the tagger cannot run afterwards"""
        self.engine = None
        if getattr(self, "_emb", None) is not None and getattr(self._emb, "model", None) is not None:
            self._emb.model.source = None

    # ------------------------------------------------------------------ config 5: frozen stack + BiLSTM (inference)
    def _build_stack(self):
        """BiLSTM(sum of embedding widths -> hidden_size, bidirectional) + linear(2 * hidden -> T) + transitions, initialised
        like the reference's torch modules (sequence_tagger_model.py:324-357,385-388,402-410); `load_stack_state` overwrites
        them with trained values.  Feature blocks follow sorted(embedding names) -- the order forward() concatenates in (:879-891)."""
        from kbner import stack as K
        embs = list(self.embeddings.embeddings) if hasattr(self.embeddings, "embeddings") else [self.embeddings]
        self._stack_embs = sorted(embs, key=lambda e: e.name)
        blocks = [int(e.embedding_length) for e in self._stack_embs]
        H, D, T = int(self.hidden_size), sum(blocks), self.tagset_size
        g = torch.Generator().manual_seed(int(torch.initial_seed() % (2 ** 31)))
        k = 1.0 / (H ** 0.5)
        u = lambda *shape, b=k: (torch.rand(*shape, generator=g) * 2 - 1) * b   # noqa: E731
        rnn = {}
        for sfx in ("", "_reverse"):
            rnn["weight_ih_l0" + sfx], rnn["weight_hh_l0" + sfx] = u(4 * H, D), u(4 * H, H)
            rnn["bias_ih_l0" + sfx], rnn["bias_hh_l0" + sfx] = u(4 * H), u(4 * H)
        kl = 1.0 / ((2 * H) ** 0.5)
        tr = torch.randn(T, T, generator=g)
        tr[self.start_idx, :] = -1e12
        tr[:, self.stop_idx] = -1e12
        self._stack_blocks = blocks
        self.load_stack_state({"rnn": rnn, "linear.weight": u(T, 2 * H, b=kl), "linear.bias": u(T, b=kl), "transitions": tr})
        self._encoders = {}

    def load_stack_state(self, state):
        """state: {"rnn": torch.nn.LSTM state dict, "linear.weight" [T, 2H], "linear.bias" [T], "transitions" [T, T]}"""
        from kbner import stack as K
        self._stack_state = {"rnn": {k: torch.as_tensor(v).detach().float().cpu() for k, v in state["rnn"].items()},
                             "linear.weight": torch.as_tensor(state["linear.weight"]).detach().float().cpu(),
                             "linear.bias": torch.as_tensor(state["linear.bias"]).detach().float().cpu(),
                             "transitions": torch.as_tensor(state["transitions"]).detach().float().cpu()}
        st = self._stack_state
        self.stack_head = K.BiLSTMHead(st["rnn"], st["linear.weight"], st["linear.bias"], self._stack_blocks, self.hidden_size,
                                       flair.device)
        self._transitions = st["transitions"].to(flair.device).contiguous()

    def _encoder_for(self, emb):
        """frozen encoder of one TransformerWordEmbeddings on the HIP engine (weights loaded once, no gradient arenas)"""
        enc = self._encoders.get(id(emb))
        if enc is None:
            from kbner import engine as E
            hc = emb.model.config
            cfg = E.EncoderConfig(vocab_size=hc.vocab_size, hidden_size=hc.hidden_size, num_hidden_layers=hc.num_hidden_layers,
                                  num_attention_heads=hc.num_attention_heads, intermediate_size=hc.intermediate_size,
                                  max_position_embeddings=hc.max_position_embeddings, type_vocab_size=getattr(hc, "type_vocab_size", 1),
                                  pad_token_id=getattr(hc, "pad_token_id", 1) if getattr(hc, "pad_token_id", 1) is not None else 1,
                                  layer_norm_eps=getattr(hc, "layer_norm_eps", 1e-5), hidden_dropout_prob=0.0,
                                  attention_probs_dropout_prob=0.0)
            enc = E.Tagger(cfg, self.tagset_size, self.start_idx, self.stop_idx, device=flair.device, inference=True)
            enc.load_hf_state_dict(emb.model.state_dict())
            self._encoders[id(emb)] = enc
        return enc

    def _forward_stack(self, sentences):
        """emissions f32 [B, n, T]: every selected embedding writes its column block of X, then BiLSTM + linear"""
        from kbner import batch as kb
        X, lengths, B, n = self._stack_input(sentences)
        head = self.stack_head
        feats = head.emissions(X, lengths, B, n)
        # tags / remove_x bookkeeping for _calculate_loss and _obtain_labels (no encoder batch here: a 1-column dummy)
        tags = np.zeros((B, n), np.int64)
        for b, s in enumerate(sentences):
            t = getattr(s, self.tag_type + "_tags", None)
            if t is not None:
                t = np.asarray(t)
                tags[b, :min(n, len(t))] = t[:n]
        hb = kb.assemble(np.zeros((B, 1), np.int64), np.ones((B, 1), np.int64), np.full((B, n), -1, np.int64), tags, lengths, self.x_idx)
        self._last = (hb, kb.to_device(hb, flair.device))
        self.mask = (torch.arange(n, device=feats.device)[None, :] < torch.from_numpy(lengths).to(feats.device)[:, None]).float()
        return feats

    def _stack_input(self, sentences):
        """the concatenated feature matrix X bf16 [rows, Dp] (token-major rows b*n + t; block `slot` at columns
        stack_head.cols[slot]), filled in place by the selected embeddings' device producers -> (X, lengths, B, n)"""
        from kbner import batch as kb
        from kbner import ops
        from flair.embeddings import BertEmbeddings, FlairEmbeddings, TransformerWordEmbeddings
        B = len(sentences)
        lengths = np.asarray([len(s) for s in sentences], np.int64)
        n = int(lengths.max())
        head = self.stack_head
        X = head.new_input(B, n)
        sel = self.selection if (getattr(self, "embedding_selector", False) and self.selection is not None) else None
        char_lms = []
        for slot, emb in enumerate(self._stack_embs):
            if sel is not None and float(sel[slot]) == 0.0:
                continue   # features * 0 (:889): the block stays zero, the embedding is not even computed
            if sel is not None and float(sel[slot]) != 1.0:
                raise NotImplementedError("fractional embedding selection weights are not supported (best_action is 0/1)")
            col = head.cols[slot]
            if isinstance(emb, TransformerWordEmbeddings):
                ids, am, first, lens, first_row = emb.prepare_stack_batch(sentences)
                hb = kb.assemble(ids, am, first, np.zeros(first.shape, np.int64), lens, None, first_row=first_row)
                db = kb.to_device(hb, flair.device)
                enc = self._encoder_for(emb)
                hidden = enc.encoder_forward(db["ids"], db["pos_ids"], db["maskbias"], db["R"], db["S"], need_grad=False)
                ops.gather_rows_into(hidden, db["row_idx"], X, col, enc.cfg.hidden_size)
            elif isinstance(emb, BertEmbeddings):
                ids, am, first, lens = emb.prepare_stack_batch(sentences)
                hb = kb.assemble(ids, am, first, np.zeros(first.shape, np.int64), lens, None, position_mode="absolute")
                db = kb.to_device(hb, flair.device)
                enc = self._encoder_for(emb)
                enc.encoder_forward(db["ids"], db["pos_ids"], db["maskbias"], db["R"], db["S"], need_grad=False)
                states = enc.acts(db["R"], db["S"]).x      # hidden_states[0..L] (0 = embedding output)
                Hh = enc.cfg.hidden_size
                for j, li in enumerate(emb.layer_indexes):  # concatenated in the order `layers` lists them (:2843-2858)
                    ops.gather_rows_into(states[len(states) + li], db["row_idx"], X, col + j * Hh, Hh)
            elif isinstance(emb, FlairEmbeddings):
                char_lms.append((emb, col))     # run together below: one launch per character step for all LMs of a width
            else:
                raise NotImplementedError("embedding class %s has no device producer" % type(emb).__name__)
        if char_lms:
            from kbner.stack import CharLMGroup
            by_width = {}
            for emb, col in char_lms:
                by_width.setdefault(emb.engine(flair.device).Hp, []).append((emb, col))
            for Hp, members in by_width.items():
                key = tuple(id(e) for e, _ in members)
                cache = self.__dict__.setdefault("_char_lm_groups", {})
                if key not in cache:
                    cache[key] = CharLMGroup([e.engine(flair.device) for e, _ in members])
                batches = [e.char_batch(sentences, n) for e, _ in members]
                cache[key].run([b[0] for b in batches], [b[1] for b in batches], X, [c for _, c in members])
        return X, lengths, B, n

    def train(self, mode: bool = True):
        """model.train() (finetune_trainer.py:938) switches the HF dropout sites and the tagger's WordDropout on; eval() off"""
        super().train(mode)
        if getattr(self, "use_rnn", False) and mode:
            log.warning("the BiLSTM / stacked tagger is inference-only here: train() mode has no effect")
        if getattr(self, "engine", None) is not None:
            self.engine.train(bool(mode) and bool(getattr(self._emb, "fine_tune", True)))
        return self

    @property
    def transitions(self):
        return self._transitions if self.engine is None else self.engine.arena.param("transitions")

    @property
    def linear(self):
        class _Lin:
            pass

        lin = _Lin()
        if self.engine is None:
            lin.weight, lin.bias = self._stack_state["linear.weight"], self._stack_state["linear.bias"]
        else:
            lin.weight = self.engine.arena.param("linear.weight")
            lin.bias = self.engine.arena.param("linear.bias")
        return lin

    def named_parameters(self, prefix="", recurse=True):
        """(name, tensor view) with the reference's naming: `transitions`, `linear.*`, `embeddings.list_embedding_0.model.<hf>`"""
        if self.engine is None:
            yield "transitions", self._transitions
            yield "linear.weight", self._stack_state["linear.weight"]
            yield "linear.bias", self._stack_state["linear.bias"]
            for k, v in self._stack_state["rnn"].items():
                yield "rnn." + k, v
            return
        if self.use_crf:   # (the reference creates `transitions` only with use_crf, :390: a softmax student has no such parameter)
            yield "transitions", self.engine.arena.param("transitions")
        yield "linear.weight", self.engine.arena.param("linear.weight")
        yield "linear.bias", self.engine.arena.param("linear.bias")
        for k, v in self.engine.hf_state_dict().items():
            yield "embeddings.list_embedding_0.model." + k, v

    def parameters(self, recurse=True):
        for _, p in self.named_parameters():
            yield p

    def zero_grad(self, set_to_none=False):
        if self.engine is not None and self.engine.arena.g is not None:    # None: a teacher after drop_gradients()
            self.engine.arena.g.zero_()
            self.engine.arena.wgrad_stale = False

    def to(self, *a, **k):
        return self

    # ------------------------------------------------------------------ batches
    def _device_batch(self, sentences):
        from kbner import batch as kb
        name = self._emb.name
        feats = getattr(sentences, "features", None)
        if feats is not None and name in feats and isinstance(feats[name], tuple):
            ids, am, first, lengths, first_row = feats[name]
        else:
            ids, am, first, lengths, first_row = self._emb.prepare_batch(sentences)
        n = first.shape[1]
        tags = np.zeros((len(sentences), n), np.int64)
        for b, s in enumerate(sentences):
            t = getattr(s, self.tag_type + "_tags", None)
            if t is not None:
                t = np.asarray(t)
                tags[b, :min(n, len(t))] = t[:n]
            else:
                tags[b, :len(s)] = [self.tag_dictionary.get_idx_for_item(tok.get_tag(self.tag_type).value) for tok in s]
        hb = kb.assemble(ids, am, first, tags, lengths, self.x_idx, first_row=first_row)
        return hb, kb.to_device(hb, flair.device)

    # ------------------------------------------------------------------ forward / loss
    def forward(self, sentences, prediction_mode=False):
        """emissions f32 [B, n, T] for every word token (device tensor); sets self.mask to the length mask"""
        if self.use_rnn:
            return self._forward_stack(sentences)
        if self.engine is None:
            raise RuntimeError("this tagger's device memory was released (release_device_memory()): it cannot run any more")
        self.embeddings.embed(sentences)
        hb, db = self._device_batch(sentences)
        feats = self.engine.forward_features(db)
        n = feats.shape[1]
        self.mask = (torch.arange(n, device=feats.device)[None, :] < db["lengths"][:, None]).float()
        self._last = (hb, db)
        return feats

    def forward_loss(self, data_points, sort=True, return_features=False):
        if isinstance(data_points, Sentence):
            data_points = [data_points]
        if self.use_rnn:
            feats = self._forward_stack(data_points)
            return self._calculate_loss(feats, data_points, self.mask)
        self.embeddings.embed(data_points)
        hb, db = self._device_batch(data_points)
        self._last = (hb, db)
        self.mask = torch.from_numpy(hb["keep"].astype(np.float32)).to(flair.device)
        return self.engine.forward_loss(db, backward=False, weights=None if self.use_crf else self._softmax_weights(hb))

    def forward_backward(self, data_points, loss_scale=1.0, sentence_weights=None, grad_ready=None, multi_view=None,
                         distill_interpolation=None):
        """forward_loss + backward into the gradient arena (what `loss.backward()` does at finetune_trainer.py:957).
        sentence_weights (optional, one per sentence) replace the 1/B of the batch mean: see ModelFinetuner.train's
        accumulation-group fusion.  grad_ready(lo, hi): called as soon as arena.g[lo:hi] is final (data-parallel overlap).

        multi_view = (indices, weights): the second half of a multi-view step (finetune_trainer.py:959-966 ->
        multi_view_loss -> _calculate_multi_view_loss, sequence_tagger_model.py:1923,1958-2093): for the sentences
        data_points[indices] (those with an `orig_sent` and S-X context) the bare sentence is encoded as a second batch and the
        tempered posteriors of its CRF are pulled towards those of the context view just computed (constant), each sentence
        weighted by `weights`.  Returns the NLL term + the distillation term (both weighted as given) as one 0-d tensor."""
        if self.use_rnn:
            raise NotImplementedError("training the BiLSTM / stacked-embedding tagger (ACE) is out of scope: config 5 is inference-only")
        self.embeddings.embed(data_points)
        hb, db = self._device_batch(data_points)
        self._last = (hb, db)
        self.mask = torch.from_numpy(hb["keep"].astype(np.float32)).to(flair.device)
        if not self.use_crf:
            if distill_interpolation is not None or multi_view:
                raise NotImplementedError("knowledge distillation / multi-view training of a softmax student (use_crf=False)")
            if sentence_weights is not None and not self.sentence_level_loss:
                raise NotImplementedError("a softmax student without sentence_loss normalises by the batch's token count: the "
                                          "accumulation-group fusion's per-sentence weights do not express that (train with "
                                          "fuse_accumulation=False)")
            w = sentence_weights if sentence_weights is not None else self._softmax_weights(hb)
            loss = self.engine.forward_loss(db, loss_scale=loss_scale, backward=True, weights=w, grad_ready=grad_ready)
            self.last_loss_parts = (loss, None)
            return loss
        if distill_interpolation is not None:
            # distill_mode (finetune_trainer.py:897-904 -> simple_forward_distillation_loss): the sentences carry teacher targets
            loss = self.engine.kd_loss(db, self._kd_batch(data_points, hb), float(distill_interpolation), float(self.temperature),
                                       loss_scale=loss_scale, backward=True, weights=sentence_weights, grad_ready=grad_ready)
            self.last_loss_parts = self.engine.last_kd_parts   # (KD terms, gold NLL), before the interpolation
            return loss
        if not multi_view or len(multi_view[0]) == 0:
            loss = self.engine.forward_loss(db, loss_scale=loss_scale, backward=True, weights=sentence_weights, grad_ready=grad_ready)
            self.last_loss_parts = (loss, None)
            return loss
        idx, kw = multi_view
        loss = self.engine.forward_loss(db, loss_scale=loss_scale, backward=True, weights=sentence_weights, grad_ready=None)
        teacher = self.engine.last_emissions.index_select(0, torch.as_tensor(list(idx), dtype=torch.long, device=flair.device))
        t_pooled = self.engine.last_pooled      # the context view's token representations (calculate_l2_loss); a constant
        t_lens = hb["clens"][list(idx)]
        orig = [data_points[i].orig_sent for i in idx]
        self.embeddings.embed(orig)
        ohb, odb = self._device_batch(orig)
        if not np.array_equal(ohb["clens"], t_lens):
            # the reference asserts the same thing (`assert (new_mask == self.mask).all()`, :2062): the context file's sentence
            # part and the plain file's sentence must be the same tokens
            raise ValueError("multi-view pair mismatch: real-token counts %s (context view) vs %s (orig_sent)" %
                             (t_lens.tolist(), ohb["clens"].tolist()))
        sel = torch.as_tensor(list(idx), dtype=torch.long, device=flair.device)
        kd = self.engine.distill_loss(odb, teacher, float(self.temperature), loss_scale=loss_scale, backward=True,
                                      weights=kw, grad_ready=grad_ready, mode="exact" if self.distill_exact else "posterior",
                                      teacher_pooled=t_pooled.index_select(0, sel) if self.calculate_l2_loss else None,
                                      l2_only=self.l2_loss_only)
        store_embeddings(orig, "none")
        self.last_loss_parts = (loss, kd)   # (weighted NLL of the context view, weighted distillation term): 0-d device tensors
        return loss + kd

    # ------------------------------------------------------------------ teacher-student knowledge distillation
    def _kd_suppress(self):
        """the tags whose teacher logits are lowered by 1e12 (finetune_trainer.py:1627-1629): STOP, START, '<unk>' -- index 0
        when the dictionary has no '<unk>' item, exactly as Dictionary.get_idx_for_item answers there"""
        td = self.tag_dictionary
        return (td.get_idx_for_item(STOP_TAG), td.get_idx_for_item(START_TAG), td.get_idx_for_item("<unk>"))

    def forward_backward_score(self, feats, lengths):
        """TEACHER side of distill_posterior: `(forward_var + backward_var) * mask` of this model's CRF over feats f32[B,n,T] with
        the START / STOP / <unk> columns lowered by 1e12 (finetune_trainer.py:1627-1634 calling _forward_alg(distill_mode=True),
        :1329-1380, and _backward_alg, :1396-1470) -> f32[B,n,T] on the device"""
        from kbner import ops
        lens = torch.as_tensor(lengths).to(device=feats.device, dtype=torch.int32).contiguous()
        return ops.crf_fb_score(feats.contiguous().float(), self.transitions, lens, self.start_idx, self.stop_idx, self._kd_suppress())

    def pair_posterior(self, feats, lengths, temperature):
        """TEACHER side of distill_exact (finetune_trainer.py:1705-1722,1885) -> (pair f32[B,n-1,T*T] softmax over (to, from),
        start_score f32[B,T], end_score f32[B,T])"""
        from kbner import ops
        lens = torch.as_tensor(lengths).to(device=feats.device, dtype=torch.int32).contiguous()
        return ops.crf_pair_posterior(feats.contiguous().float(), self.transitions, lens, float(temperature), self.start_idx,
                                      self.stop_idx, self._kd_suppress())

    def _kd_batch(self, sentences, hb):
        """the batch's teacher targets (Sentence.set_teacher_*; finetune_trainer.py:1866-1886) padded to the batch's token count
        and moved to the device -- what `resort` (:1911-2060) and the `get_teacher_*` stacks of simple_forward_distillation_loss
        (:2128-2131,2253-2256,2283-2286) do in the reference -> the `kd` dict of kbner.engine.Tagger.kd_loss"""
        B, T = len(sentences), self.tagset_size
        n = hb["row_idx"].size // B
        dev = flair.device
        kd = {}

        def teachers_of(attr):
            k = {len(getattr(sn, attr)) for sn in sentences}
            if len(k) != 1 or 0 in k:
                raise ValueError("every sentence of a distill_mode batch needs the same, non-zero number of teacher entries in %s "
                                 "(got %s): was ModelFinetuner.assign_pretrained_teacher_targets run on this data, and does "
                                 "every training corpus have a teacher?" % (attr, sorted(k)))
            return k.pop()

        if self.distill_posterior:
            nt = teachers_of("_teacher_posteriors")
            sc = np.zeros((nt, B, n, T), np.float32)
            for b, sn in enumerate(sentences):
                for t, p in enumerate(sn._teacher_posteriors):
                    L = min(len(sn), n)
                    sc[t, b, :L] = p[:L]
            kd["scores"] = [torch.from_numpy(sc[t]).to(dev) for t in range(nt)]
        if self.distill_exact:
            teachers_of("_teacher_posteriors")
            pair = np.zeros((B, max(n - 1, 0), T * T), np.float32)
            s_sc, e_sc = np.zeros((B, T), np.float32), np.zeros((B, T), np.float32)
            for b, sn in enumerate(sentences):      # the loss reads teacher 0 only (`[:, :, 0]`, :2163-2166)
                p = sn._teacher_posteriors[0]
                L = min(len(p), max(n - 1, 0))
                pair[b, :L] = p[:L]
                s_sc[b], e_sc[b] = sn._teacher_startscores[0], sn._teacher_endscores[0]
            kd["exact"] = (torch.from_numpy(pair).to(dev), torch.from_numpy(s_sc).to(dev), torch.from_numpy(e_sc).to(dev))
        if self.distill_emission:
            if self.distill_posterior:      # :2349-2353: the first teacher's forward-backward scores
                kd["emission"] = (kd["scores"][0], self.distill_prob)
            else:                           # :2355-2358: mean over the teachers of the stored predictions, zero rows behind the end
                teachers_of("_teacher_prediction")
                pr = np.zeros((B, n, T), np.float32)
                for b, sn in enumerate(sentences):
                    p = sn.get_teacher_prediction()
                    L = min(len(p), n)
                    pr[b, :L] = p[:L]
                kd["emission"] = (torch.from_numpy(pr).to(dev), self.distill_prob)
        if self.distill_crf:
            nt = teachers_of("_teacher_target")
            tg = [sn.get_teacher_target() for sn in sentences]          # [len, best_k * teachers]
            K = tg[0].shape[1]
            targets = np.zeros((B, n, K), np.int32)
            for b, t in enumerate(tg):
                L = min(len(t), n)
                targets[b, :L] = t[:L]
            kd["targets"] = torch.from_numpy(targets).to(dev)
            if self.crf_attention:
                teachers_of("_teacher_weights")
                att = np.stack([sn.get_teacher_weights() for sn in sentences], 0).astype(np.float32)      # [B, K]
                att_nums = sum(len(sn._teacher_weights) for sn in sentences)
                if self.distill_with_gold:
                    # :2289-2304: paths that disagree with the gold tags on more tokens weigh less
                    valid = np.arange(n)[None, :] < np.asarray([len(sn) for sn in sentences])[:, None]
                    num_error = ((targets != hb["tags"][:, :, None]) & valid[:, :, None]).sum(1).astype(np.float32)   # [B, K]
                    g = float(self.gold_const)
                    score_w = np.exp(-num_error / g) if self.exp_score else g / (num_error + g)
                    att = att * score_w
                    att = att / att.sum(-1, keepdims=True) * (att_nums / float(B))
                kd["weights"] = torch.from_numpy(np.ascontiguousarray(att, np.float32)).to(dev)
                kd["att_nums"] = att_nums
        return kd

    def simple_forward_distillation_loss(self, data_points, teacher_data_points=None, teacher=None, sort=True, interpolation=0.5,
                                         train_with_professor=False, professor_interpolation=0.5, language_attention_warmup=False,
                                         calc_teachers_target_loss=False, language_weight=None, biaffine=None, language_vector=None):
        """sequence_tagger_model.py:2110-2372 for a CRF student: interpolation * (posterior + crf + exact KD terms) +
        (1 - interpolation) * gold NLL, as a 0-d device tensor (forward only; training goes through forward_backward(...,
        distill_interpolation=...) since there is no autograd graph).  The sentences must carry teacher targets."""
        if teacher is not None or train_with_professor or calc_teachers_target_loss or language_attention_warmup:
            raise NotImplementedError("online teachers / professors / language attention are outside the hot path")
        if isinstance(data_points, Sentence):
            data_points = [data_points]
        self.embeddings.embed(data_points)
        hb, db = self._device_batch(data_points)
        self._last = (hb, db)
        self.mask = torch.from_numpy(hb["keep"].astype(np.float32)).to(flair.device)
        loss = self.engine.kd_loss(db, self._kd_batch(data_points, hb), float(interpolation), float(self.temperature), backward=False)
        self.last_loss_parts = self.engine.last_kd_parts
        return loss

    def check_multi_view(self, sentences):
        """sequence_tagger_model.py:1928-1956: False unless some sentence of the batch carries an `orig_sent` and the batch's tags
        contain S-X; otherwise the [B, n] tag tensor"""
        if sum(hasattr(s, "orig_sent") for s in sentences) == 0:
            return False
        n = max(len(s) for s in sentences)
        tags = np.zeros((len(sentences), n), np.int64)
        for b, s in enumerate(sentences):
            tags[b, :len(s)] = [self.tag_dictionary.get_idx_for_item(t.get_tag(self.tag_type).value) for t in s]
        x = self.tag_dictionary.get_idx_for_item("S-X")
        if not (tags == x).any():
            return False
        return torch.from_numpy(tags)

    def multi_view_plan(self, sentences, with_flag=False):
        """-> indices of the sentences that take part in the distillation term of this (micro-)batch: those with an `orig_sent`
        whose own tags contain S-X (:2023); [] when check_multi_view is False (the batch is then a plain NLL batch).
        with_flag: -> (is a multi-view batch, indices) from ONE walk over the batch's tags (the trainer's group weights need both)"""
        tags = self.check_multi_view(sentences)
        if tags is False:
            return (False, []) if with_flag else []
        x = self.tag_dictionary.get_idx_for_item("S-X")
        sel = [b for b, s in enumerate(sentences) if hasattr(s, "orig_sent") and bool((tags[b, :len(s)] == x).any())]
        return (True, sel) if with_flag else sel

    def touched_word_ids(self, sentences):
        """the word-embedding rows a list of sentences looks up (host integers): what the data-parallel gradient exchange
        needs to send only the touched rows of the [V, H] embedding gradient"""
        ids = self._emb.prepare_batch(sentences)[0]
        # + the ids the engine pads with (0 inside a row like the reference, the pad id for the rows up to a multiple of 256)
        return np.union1d(np.asarray(ids).ravel(), np.asarray([0, self.engine.cfg.pad_token_id]))

    def _softmax_weights(self, hb):
        """per-sentence weights of the softmax head's loss: 1 / B (sentence_loss) or 1 / (kept tokens of the batch) (:2534-2539)"""
        B = hb["clens"].shape[0]
        denom = float(B) if self.sentence_level_loss else float(max(1, int(hb["clens"].sum())))
        return torch.full((B,), 1.0 / denom, dtype=torch.float32, device=flair.device)

    def _calculate_loss(self, features, sentences, mask):
        """CRF NLL of given emissions [B,n,T] (mean over sentences); narrows self.mask to the non-S-X tokens like the
        reference's remove_x branch does (:2448-2453)."""
        from kbner import ops
        hb, db = self._last if getattr(self, "_last", None) is not None else self._device_batch(sentences)
        B, n, T = features.shape
        flat = features.reshape(B * n, T).contiguous()
        nc = hb["ctags"].shape[1]
        # (compaction index and keep mask travelled with the batch: no host->device copy here, which would block the host until
        # the forward pass has drained and serialise evaluate()'s host / device pipeline)
        gathered = ops.gather_rows_f32(flat, db["cfeat_idx"])
        if not self.use_crf:
            # :2523-2539: token-level cross entropy under the narrowed mask, summed; / B with sentence_loss, / mask.sum() without
            w = self._softmax_weights(hb)
            per, _ = ops.softmax_ce(gathered.view(B, nc, T).contiguous(), db["ctags"], db["clens"], w)
            self.mask = db["keep_f"]
            self._last_compact = gathered.view(B, nc, T)
            return (per * w).sum()
        logz, gold, _ = ops.crf_nll_fwd(gathered.view(B, nc, T).contiguous(), self.transitions, db["ctags"],
                                        db["clens"], self.start_idx, self.stop_idx)
        self.mask = db["keep_f"]
        self._last_compact = gathered.view(B, nc, T)
        return (logz - gold).mean()

    def _obtain_labels(self, feature, sentences, get_all_tags: bool = False):
        """Viterbi over the rows self.mask keeps, S-X / confidence 1 re-padded around them (:1193-1210)."""
        from kbner import ops
        B, n, T = feature.shape
        if not self.use_crf:
            # :1177-1180, 1212-1218: arg-max of the emissions at EVERY token of the sentence (no S-X compaction on this branch),
            # confidence = the softmax probability of that tag; get_all_tags: the whole distribution per token
            full = torch.tensor([len(s) for s in sentences], dtype=torch.int32, device=feature.device)
            res = ops.softmax_decode(feature.contiguous().float(), full, want_dist=bool(get_all_tags))
            tg, cf = res[0].cpu().numpy(), res[1].cpu().numpy()
            dist = res[2].cpu().numpy() if get_all_tags else None
            out, all_tags = [], []
            for b, s in enumerate(sentences):
                out.append([Label(self.tag_dictionary.get_item_for_index(int(tg[b, i])), float(cf[b, i])) for i in range(len(s))])
                if get_all_tags:
                    all_tags.append([[Label(self.tag_dictionary.get_item_for_index(k), float(dist[b, i, k])) for k in range(T)]
                                     for i in range(len(s))])
            return out, all_tags
        keep = self.mask.bool().cpu().numpy() if self.mask is not None else np.ones((B, n), bool)
        if getattr(self, "predict_posterior", False):
            # :1182-1192,1212-1218: marginals softmax(alpha + beta) over the WHOLE token sequence (no S-X compaction);
            # positions self.mask zeroes get the all-zero score row, i.e. a uniform distribution and tag index 0
            full = torch.tensor([len(s) for s in sentences], dtype=torch.int32, device=feature.device)
            marg = ops.crf_posterior(feature.contiguous(), self.transitions, full, self.start_idx,
                                     self.stop_idx).cpu().numpy()
            out = []
            for b, s in enumerate(sentences):
                row = []
                for i in range(len(s)):
                    if keep[b, i]:
                        k = int(marg[b, i].argmax())
                        row.append(Label(self.tag_dictionary.get_item_for_index(k), float(marg[b, i, k])))
                    else:
                        row.append(Label(self.tag_dictionary.get_item_for_index(0), 1.0 / T))
                out.append(row)
            return out, []
        lens = keep.sum(1).astype(np.int32)
        nc = max(1, int(lens.max()))
        idx = torch.from_numpy(_compact_index(keep, nc)).to(feature.device)
        comp = ops.gather_rows_f32(feature.reshape(B * n, T).contiguous(), idx)
        tags, conf = ops.crf_viterbi(comp.view(B, nc, T).contiguous(), self.transitions,
                                     torch.from_numpy(lens).to(feature.device), self.start_idx, self.stop_idx)
        tags, conf = tags.cpu().numpy(), conf.cpu().numpy()
        x_item = "S-X"
        out = []
        for b, s in enumerate(sentences):
            L, k = len(s), int(lens[b])
            path = [self.tag_dictionary.get_item_for_index(int(t)) for t in tags[b, :k]]
            cf = [float(c) for c in conf[b, :k]]
            if k < L:
                kept = np.nonzero(keep[b])[0]
                before = int(kept[0]) if len(kept) else 0
                path = [x_item] * before + path
                cf = [1] * before + cf          # the reference pads with the INTEGER 1 (:1204-1207): prediction files print "1"
                after = L - len(path)
                path, cf = path + [x_item] * after, cf + [1] * after
            out.append([Label(p, c) for p, c in zip(path, cf)])
        return out, []

    def _viterbi_decode_nbest(self, feats, mask, nbest):
        """(path_score [B, nbest], decode_idx [B, n, nbest]) -- the NCRF++ n-best decoder of sequence_tagger_model.py:1660, as the
        KD trainers call it on a teacher (finetune_trainer.py:1600): HIP kernel, conventions and quirks of the reference kept"""
        from kbner import ops
        lens = mask.to(feats.device).long().sum(1).to(torch.int32).contiguous()
        score, dec = ops.crf_viterbi_nbest(feats.contiguous().float(), self.transitions, lens, self.start_idx, self.stop_idx, int(nbest))
        return score, dec.long()

    def _gold_x_token_ids(self, sentence):
        """1-based ids of the tokens whose gold tag is exactly 'S-X' (:2663; from the loader's tag-id row when present)"""
        row = getattr(sentence, self.tag_type + "_tags", None)
        sx = self.tag_dictionary.get_idx_for_item("S-X") if "S-X" in self.tag_dictionary.get_items() else None
        if row is not None and sx is not None:
            r = np.asarray(row)[:len(sentence)]
            return set((np.nonzero(r == sx)[0] + 1).tolist())
        return {t.idx for t in sentence.tokens if t.get_tag(self.tag_type).value == "S-X"}

    # ------------------------------------------------------------------ evaluation
    def evaluate(self, data_loader, out_path: Path = None, embeddings_storage_mode: str = "cpu", prediction_mode=False,
                 speed_test=False, shard=None):
        """sequence_tagger_model.py:2593-2729.  Under speed_test only forward + decode run (no loss, so self.mask keeps ALL
        tokens and Viterbi decodes the context too; no prediction lines, no metric) -- as in the reference.

        shard = (rank, world_size): data-parallel evaluation -- this rank takes batches rank, rank + world, ...; the metric's
        tp / fp / fn / tn counters, the loss sum and the batch count are summed over ranks (kbner.dp) before the Result is
        built, so EVERY rank returns the Result one rank scoring all batches would return (the loss as the mean over all
        batches).  Not combined with out_path (the prediction file is written by one rank over the whole set).

        Host / device pipeline: batch k+1's encoder + loss + Viterbi are ENQUEUED (with the device->host copy of its tags into
        pinned memory) before batch k's labels, prediction lines, spans and metric are built on the host, so the two overlap;
        spans come from tag-id arrays (flair.data.batch_spans), not from per-token Label objects -- a KB-NER sentence is
        ~95 % single-token S-X context that the remove_x rule discards anyway."""
        import time
        from flair.data import span_tables
        eval_loss, batch_no = 0.0, 0
        metric = Metric("Evaluation")
        if shard is not None and (out_path is not None or shard[1] <= 1):
            if out_path is not None and shard[1] > 1:
                raise ValueError("evaluate(shard=...) cannot write a prediction file: evaluate the whole set on one rank")
            shard = None
        outfile = open(out_path, "w", encoding="utf-8") if out_path is not None else None
        env = {"items": self.tag_dictionary.get_items(), "metric": metric, "outfile": outfile, "speed_test": speed_test}
        env["tables"] = span_tables(env["items"])
        env["x_id"] = env["items"].index("S-X") if "S-X" in env["items"] else None
        env["unk_id"] = env["items"].index("<unk>") if "<unk>" in env["items"] else None
        t0 = time.time()
        failure = None
        try:
            pending = None
            for index, batch in enumerate(data_loader):
                if shard is not None and index % shard[1] != shard[0]:
                    continue
                batch_no += 1
                st = self._eval_enqueue(batch, prediction_mode, speed_test)
                if pending is not None:
                    eval_loss += self._eval_finish(pending[0], pending[1], env)
                    store_embeddings(pending[0], embeddings_storage_mode)
                pending = (batch, st)
            if pending is not None:
                eval_loss += self._eval_finish(pending[0], pending[1], env)
                store_embeddings(pending[0], embeddings_storage_mode)
        except BaseException as e:   # (also KeyboardInterrupt) a sharded evaluation must still reach the collective below, or the
            if shard is None:        # other ranks block in it forever; the failure is re-raised -- on every rank -- after it
                raise
            failure = e
        finally:
            if outfile is not None:
                outfile.close()
        if speed_test:
            rate = getattr(data_loader, "num_examples", 0) / max(1e-9, time.time() - t0)
            print(rate)
            log.info("decode speed: %.2f sents/sec", rate)
        if shard is not None:
            from kbner import dp
            parts = dp.all_gather_object((metric.counts(), eval_loss, batch_no, None if failure is None else repr(failure)))
            if failure is not None:
                raise failure
            failed = [(r, p[3]) for r, p in enumerate(parts) if p[3] is not None]
            if failed:
                raise RuntimeError("evaluate(): rank(s) %s failed" % failed)
            metric = Metric("Evaluation")
            eval_loss, batch_no = 0.0, 0
            for counts, loss_sum, nb, _ in parts:   # rank order: the same sums on every rank
                metric.merge_counts(counts)
                eval_loss += loss_sum
                batch_no += nb
        eval_loss /= max(1, batch_no)
        detailed = ("\nMICRO_AVG: acc {} - f1-score {}\nMACRO_AVG: acc {} - f1-score {}".format(
            metric.micro_avg_accuracy(), metric.micro_avg_f_score(), metric.macro_avg_accuracy(), metric.macro_avg_f_score()))
        for c in metric.get_classes():
            detailed += ("\n{:<10} tp: {} - fp: {} - fn: {} - tn: {} - precision: {:.4f} - recall: {:.4f} - accuracy: {:.4f} - "
                         "f1-score: {:.4f}".format(c, metric.get_tp(c), metric.get_fp(c), metric.get_fn(c), metric.get_tn(c),
                                                   metric.precision(c), metric.recall(c), metric.accuracy(c), metric.f_score(c)))
        result = Result(main_score=metric.micro_avg_f_score(), log_line="{}\t{}\t{}".format(metric.precision(), metric.recall(),
                                                                                           metric.micro_avg_f_score()),
                        log_header="PRECISION\tRECALL\tF1", detailed_results=detailed, macro_score=metric.macro_avg_f_score())
        return result, eval_loss

    def _eval_enqueue(self, batch, prediction_mode, speed_test):
        """device half of one evaluate() batch: everything is enqueued, nothing is waited for"""
        from kbner import ops
        features = self.forward(batch, prediction_mode=prediction_mode)
        last = getattr(self, "_last", None)
        if (last is None or features is None or getattr(self, "use_rnn", False) or getattr(self, "predict_posterior", False)
                or not getattr(self, "use_crf", True)):
            # posterior decoding, the stacked-embedding tagger (and test doubles that replace forward / _obtain_labels): the
            # generic synchronous path, labels -> per-token spans
            loss = None if speed_test else self._calculate_loss(features, batch, self.mask)
            return {"labels": self._obtain_labels(features, batch)[0], "loss": loss}
        hb, db = last
        B, n, T = features.shape
        loss = None
        if not speed_test:
            loss = self._calculate_loss(features, batch, self.mask)       # narrows self.mask to the non-S-X tokens
            keep = hb["keep"]
            lens, nc = hb["clens"], hb["ctags"].shape[1]
            comp, lens_d = self._last_compact.contiguous(), db["clens"]   # the kept rows the loss just gathered
        else:
            keep = np.arange(n)[None, :] < np.asarray([len(s) for s in batch])[:, None]
            lens, nc = keep.sum(1).astype(np.int32), n
            comp, lens_d = features.contiguous(), db["lengths"]           # every token is decoded (:2612-2622)
        tags, conf = ops.crf_viterbi(comp, self.transitions, lens_d, self.start_idx, self.stop_idx)
        tags_h = torch.empty(tags.shape, dtype=tags.dtype, pin_memory=True)
        conf_h = torch.empty(conf.shape, dtype=conf.dtype, pin_memory=True)
        tags_h.copy_(tags, non_blocking=True)
        conf_h.copy_(conf, non_blocking=True)
        loss_h = None
        if loss is not None:
            loss_h = torch.empty((), dtype=torch.float32, pin_memory=True)
            loss_h.copy_(loss.detach().float().reshape(()), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return {"tags": tags_h, "conf": conf_h, "loss_h": loss_h, "event": ev, "keep": keep, "lens": lens, "hb": hb, "nc": nc}

    def _eval_finish(self, batch, st, env):
        """host half: labels on every token, prediction lines, spans, metric.  Returns this batch's loss (0.0 under speed_test)."""
        from flair.data import batch_spans
        items, outfile = env["items"], env["outfile"]
        loss_val = 0.0
        fast = False
        if "labels" in st:       # labels were built synchronously (see _eval_enqueue)
            all_labels = st["labels"]
            if st["loss"] is not None:
                loss_val = float(st["loss"])
        else:
            keep, hb = st["keep"], st["hb"]
            B, n = keep.shape
            st["event"].synchronize()
            tags, conf, lens = st["tags"].numpy(), st["conf"].numpy(), st["lens"]
            if st["loss_h"] is not None:
                loss_val = float(st["loss_h"])
            x_id = env["x_id"] if env["x_id"] is not None else 0
            # the reference re-pads around the decoded tags with S-X / the INTEGER confidence 1: `before` = the first kept
            # position, the k decoded tags follow contiguously, S-X up to the sentence length (:1202-1208)
            first = np.where(keep.any(1), keep.argmax(1), 0)
            col = np.arange(n)[None, :]
            sel = (col >= first[:, None]) & (col < (first + lens)[:, None])
            valid_c = np.arange(st["nc"])[None, :] < lens[:, None]
            pred_ids = np.full((B, n), x_id, np.int64)
            pred_sc = np.ones((B, n), np.float32)
            pred_ids[sel] = tags[valid_c]
            pred_sc[sel] = conf[valid_c]
            x_label = Label("S-X", 1)      # ONE object for every padded token of the batch (a Label is a value)
            all_labels = []
            for b, s in enumerate(batch):
                L, k, f = len(s), int(lens[b]), int(first[b])
                if k >= L:
                    row = [Label(items[t], float(c)) for t, c in zip(tags[b, :k], conf[b, :k])]
                else:
                    row = [x_label] * f + [Label(items[t], float(c)) for t, c in zip(tags[b, :k], conf[b, :k])] + \
                          [x_label] * (L - f - k)
                all_labels.append(row)
        if env["speed_test"]:
            return 0.0
        tag_type = self.tag_type
        for sentence, sent_tags in zip(batch, all_labels):
            for token, tag in zip(sentence.tokens, sent_tags):
                token.add_tag_label("predicted", tag)
            if outfile is not None:
                outfile.write("".join("{} {} {} {}\n".format(token.text, token.get_tag(tag_type).value, tag.value, tag.score)
                                      for token, tag in zip(sentence.tokens, sent_tags)) + "\n")
        if "labels" not in st:
            gold_ids = hb["tags"]
            # (a gold tag missing from the dictionary maps to <unk> in the id row: then the per-token string path below decides)
            in_len = np.arange(n)[None, :] < np.asarray([len(s) for s in batch])[:, None]
            fast = gold_ids.shape == keep.shape and not (env["unk_id"] is not None and
                                                         bool(((gold_ids == env["unk_id"]) & in_len).any()))
        if fast:
            if self.remove_x:
                # :2653-2672: a predicted span is dropped iff it contains a token whose GOLD tag is exactly 'S-X' (predicted
                # X-class spans on other tokens stay and count as false positives); gold spans of class X are dropped
                gx = (gold_ids == env["x_id"]) if env["x_id"] is not None else np.zeros_like(keep)
                gold_sp = batch_spans(batch, gold_ids, None, env["tables"], skip_class="X")
                pred_sp = batch_spans(batch, pred_ids, pred_sc, env["tables"], drop_flags=gx)
            else:
                gold_sp = batch_spans(batch, gold_ids, None, env["tables"])
                pred_sp = batch_spans(batch, pred_ids, pred_sc, env["tables"])
        metric = env["metric"]
        for b, sentence in enumerate(batch):
            if fast:
                gold = [(s.tag, str(s)) for s in gold_sp[b]]
                pred = [(s.tag, str(s)) for s in pred_sp[b]]
            elif self.remove_x:
                gold = [(s.tag, str(s)) for s in sentence.get_spans(tag_type, skip_class="X")]
                pred = [(s.tag, str(s)) for s in sentence.get_spans("predicted", drop_touching=self._gold_x_token_ids(sentence))]
            else:
                gold = [(s.tag, str(s)) for s in sentence.get_spans(tag_type)]
                pred = [(s.tag, str(s)) for s in sentence.get_spans("predicted")]
            for tag, span in pred:
                (metric.add_tp if (tag, span) in gold else metric.add_fp)(tag)
            for tag, span in gold:
                (metric.add_fn if (tag, span) not in pred else metric.add_tn)(tag)
        return loss_val

    def predict(self, sentences, mini_batch_size=32, embedding_storage_mode="none"):
        from flair.custom_data_loader import BatchedData
        if isinstance(sentences, Sentence):
            sentences = [sentences]
        for i in range(0, len(sentences), mini_batch_size):
            batch = BatchedData(sentences[i:i + mini_batch_size])
            feats = self.forward(batch)
            tags, _ = self._obtain_labels(feats, batch)
            for s, ts in zip(batch, tags):
                for tok, tag in zip(s.tokens, ts):
                    tok.add_tag_label(self.tag_type, tag)
        return sentences

    # ------------------------------------------------------------------ persistence
    def _get_state_dict(self):
        return {
            "format": "kbner-mi355x-v1",
            "encoder_state_dict": {k: v.detach().cpu() for k, v in self.engine.hf_state_dict().items()},
            "linear.weight": self.engine.arena.param("linear.weight").detach().cpu().clone(),
            "linear.bias": self.engine.arena.param("linear.bias").detach().cpu().clone(),
            "transitions": self.engine.arena.param("transitions").detach().cpu().clone(),
            "tag_dictionary": self.tag_dictionary, "tag_type": self.tag_type, "hidden_size": self.hidden_size,
            "use_crf": self.use_crf, "use_rnn": False, "remove_x": self.remove_x, "sentence_loss": self.sentence_level_loss,
            "word_dropout": self.use_word_dropout, "embedding_model_dir": getattr(self._emb, "name", None),
            "trained_epochs": self.trained_epochs,
            # loss switches, so that a model re-loaded as a student (load_pretrained) keeps training the way it was configured
            "temperature": self.temperature, "multi_view_training": self.multi_view_training,
            "distill_posterior": self.distill_posterior, "distill_crf": self.distill_crf, "distill_exact": self.distill_exact,
            "crf_attention": self.crf_attention, "distill_with_gold": self.distill_with_gold, "exp_score": self.exp_score,
            "gold_const": self.gold_const, "calculate_l2_loss": self.calculate_l2_loss, "l2_loss_only": self.l2_loss_only,
        }

    @classmethod
    def _init_model_with_state_dict(cls, state):
        from flair.embeddings import StackedEmbeddings, TransformerWordEmbeddings
        emb = TransformerWordEmbeddings(model=state["embedding_model_dir"], layers="-1", pooling_operation="first", fine_tune=True)
        model = cls(hidden_size=state["hidden_size"], embeddings=StackedEmbeddings([emb]), tag_dictionary=state["tag_dictionary"],
                    tag_type=state["tag_type"], use_crf=bool(state.get("use_crf", True)), use_rnn=False, remove_x=state["remove_x"],
                    sentence_loss=state["sentence_loss"], word_dropout=state.get("word_dropout", 0.0), dropout=0.0,
                    locked_dropout=0.0, temperature=state.get("temperature", 1),
                    **{k: state.get(k, False) for k in ("multi_view_training", "distill_posterior", "distill_crf", "distill_exact",
                                                        "crf_attention", "distill_with_gold", "exp_score", "calculate_l2_loss", "l2_loss_only")},
                    gold_const=state.get("gold_const", 1.0))
        model.engine.load_hf_state_dict(state["encoder_state_dict"])
        for k in ("linear.weight", "linear.bias", "transitions"):
            model.engine.set_param(k, state[k])
        model.trained_epochs = state.get("trained_epochs", 0)
        return model


def _compact_index(keep, nc):
    """flat row index [B*nc] into [B*n] of the kept tokens, -1 padded"""
    B, n = keep.shape
    out = np.full((B, nc), -1, np.int32)
    for b in range(B):
        k = np.nonzero(keep[b])[0]
        out[b, :len(k)] = b * n + k
    return out.reshape(-1)


class FastSequenceTagger(SequenceTagger):
    """same engine; the reference's 'Fast' variant differs only in how it batches the torch ops"""
    pass
