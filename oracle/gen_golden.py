#!/usr/bin/env python
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (imported read-only from /root/reference
through oracle/ref_import.py's shims) in the build container.  Run:  python oracle/gen_golden.py

Fixtures are data only (inputs + the reference's outputs).  The generator is committed so the
vectors are reproducible; /root/reference does not exist on the GPU box and nothing there needs it.

Vectors (ids follow SURVEY.md §8c):
  G1/G2  crf_forward_score.npz  _forward_alg (:1329) + FastSequenceTagger._score_sentence (:2544)
  G3     crf_loss_grad.npz      _calculate_loss incl. remove_x (:2426) + autograd d feats / d transitions
  G4     viterbi.npz            _viterbi_decode (:1248) incl. crafted ties and large magnitudes
  G5     obtain_labels.npz      _obtain_labels remove_x re-padding (:1157)
  G6     encoder_tiny.npz       transformers 5.15 XLMRobertaModel (eager, fp32) tiny + 1-layer wide config
  G8/G9  adamw.npz              reference trainer's optimiser grouping + linear schedule over 3 steps
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

from oracle import ref_import  # noqa: E402


def build_reference_tagger(flair, T_dict_path, remove_x=True):
    from flair.data import Dictionary
    from flair.models import FastSequenceTagger

    tag_dictionary = Dictionary.load_from_file(T_dict_path)

    class _DummyEmb(torch.nn.Module):
        embedding_length = 8
        name = "dummy"
        embeddings = []

        def embed(self, *a, **k):
            pass

    emb = _DummyEmb()
    emb.embeddings = [emb]
    torch.manual_seed(1234)
    tagger = FastSequenceTagger(hidden_size=8, embeddings=emb, tag_dictionary=tag_dictionary, tag_type="ner",
                                use_crf=True, use_rnn=False, use_cnn=False, dropout=0.0, word_dropout=0.0,
                                locked_dropout=0.0, sentence_loss=True, remove_x=remove_x, config=None)
    return tagger, tag_dictionary


class _Sent:
    """Minimal stand-in for flair.data.Sentence as _calculate_loss/_obtain_labels use it:
    `len(sentence.tokens)` and `sentence.ner_tags` (custom_data_loader.py:356-374)."""

    def __init__(self, n, tags):
        self.tokens = [None] * n
        self.ner_tags = torch.as_tensor(tags, dtype=torch.int64)

    def __len__(self):
        return len(self.tokens)


def main():
    os.makedirs(GOLD, exist_ok=True)
    flair = ref_import.load_reference()
    from flair.models.sequence_tagger_model import START_TAG, STOP_TAG

    dict_path = os.path.join(ref_import.REFERENCE_ROOT, "resources/taggers/EN-English_x.pkl")
    tagger, td = build_reference_tagger(flair, dict_path)
    T = len(td)
    start = td.get_idx_for_item(START_TAG)
    stop = td.get_idx_for_item(STOP_TAG)
    x_idx = td.get_idx_for_item("S-X")
    items = [s for s in td.get_items()]
    print("T", T, "start", start, "stop", stop, "S-X", x_idx)
    rng = np.random.default_rng(20220711)

    # ---------------- G1/G2: forward score + gold score ----------------
    cases = {}
    ci = 0
    trans_init = tagger.transitions.detach().clone().numpy()
    trans_pert = trans_init.copy()
    pert = rng.standard_normal((T, T)).astype(np.float32) * 0.5
    keepm = trans_pert > -1e11
    trans_pert[keepm] += pert[keepm]
    for trans in (trans_init, trans_pert):
        for (B, n) in ((1, 1), (1, 2), (3, 7), (3, 40), (2, 5)):
            feats = (rng.standard_normal((B, n, T)) * 2.0).astype(np.float32)
            lens = rng.integers(1, n + 1, size=B)
            lens[0] = n
            if (B, n) == (2, 5):
                lens[1] = 0  # an all-context sentence after compaction
            valid = [i for i in range(T) if i not in (start, stop, x_idx, 0)]
            tags = rng.choice(valid, size=(B, n)).astype(np.int64)
            for b in range(B):
                tags[b, lens[b]:] = 0
            with torch.no_grad():
                tagger.transitions.copy_(torch.from_numpy(trans))
                lt = torch.from_numpy(lens.astype(np.int64))
                alpha = tagger._forward_alg(torch.from_numpy(feats), lt)
                mask = (torch.arange(n)[None, :] < lt[:, None]).float()
                gold = tagger._score_sentence(torch.from_numpy(feats), torch.from_numpy(tags), lt, mask=mask)
            cases["c%d_feats" % ci] = feats
            cases["c%d_lens" % ci] = lens.astype(np.int64)
            cases["c%d_tags" % ci] = tags
            cases["c%d_trans" % ci] = trans
            cases["c%d_alpha" % ci] = alpha.numpy()
            cases["c%d_gold" % ci] = gold.numpy()
            ci += 1
    cases["n_cases"] = np.int64(ci)
    cases["start"], cases["stop"], cases["x_idx"] = np.int64(start), np.int64(stop), np.int64(x_idx)
    np.savez_compressed(os.path.join(GOLD, "crf_forward_score.npz"), **cases)

    # ---------------- G3: _calculate_loss with remove_x + autograd ----------------
    cases = {}
    ci = 0
    for (B, n, nreal) in ((2, 12, (4, 7)), (3, 30, (5, 1, 9)), (1, 6, (6,)), (4, 50, (16, 3, 8, 11))):
        feats = (rng.standard_normal((B, n, T)) * 1.5).astype(np.float32)
        lengths = []
        tags = np.zeros((B, n), np.int64)
        valid = [i for i in range(T) if i not in (start, stop, x_idx, 0)]
        for b in range(B):
            L = int(rng.integers(max(nreal[b], n // 2), n + 1)) if b > 0 else n
            lengths.append(L)
            tags[b, :nreal[b]] = rng.choice(valid, size=nreal[b])
            tags[b, nreal[b]:L] = x_idx  # <EOS> + context tokens are S-X after BIOES conversion
        lengths = np.asarray(lengths, np.int64)
        ft = torch.from_numpy(feats).clone().requires_grad_(True)
        with torch.no_grad():
            tagger.transitions.copy_(torch.from_numpy(trans_pert))
        tagger.transitions.grad = None
        sents = [_Sent(int(lengths[b]), tags[b]) for b in range(B)]
        mask = (torch.arange(n)[None, :] < torch.from_numpy(lengths)[:, None]).float()
        loss = tagger._calculate_loss(ft, sents, mask)
        loss.backward()
        cases["c%d_feats" % ci] = feats
        cases["c%d_lengths" % ci] = lengths
        cases["c%d_tags" % ci] = tags
        cases["c%d_trans" % ci] = trans_pert
        cases["c%d_loss" % ci] = loss.detach().numpy()
        cases["c%d_dfeats" % ci] = ft.grad.numpy()
        cases["c%d_dtrans" % ci] = tagger.transitions.grad.detach().clone().numpy()
        cases["c%d_maskout" % ci] = tagger.mask.detach().numpy()
        ci += 1
    cases["n_cases"] = np.int64(ci)
    cases["start"], cases["stop"], cases["x_idx"] = np.int64(start), np.int64(stop), np.int64(x_idx)
    np.savez_compressed(os.path.join(GOLD, "crf_loss_grad.npz"), **cases)

    # ---------------- G4: Viterbi, bit-exact ----------------
    cases = {}
    ci = 0

    def run_vit(feats, trans):
        with torch.no_grad():
            tagger.transitions.copy_(torch.from_numpy(trans))
            conf, path, _ = tagger._viterbi_decode(torch.from_numpy(feats))
        return np.asarray([int(p) for p in path], np.int32), np.asarray(conf, np.float32)

    vit_inputs = []
    for n in (1, 2, 7, 40, 128):
        vit_inputs.append(((rng.standard_normal((n, T)) * 2).astype(np.float32), trans_init))
        vit_inputs.append(((rng.standard_normal((n, T)) * 2).astype(np.float32), trans_pert))
    # crafted ties: quantised emissions + quantised transitions => many exact ties
    tq = np.round(trans_pert * 2) / 2
    tq[start, :] = -1e12
    tq[:, stop] = -1e12
    tq = tq.astype(np.float32)
    for n in (5, 33):
        vit_inputs.append((np.round(rng.standard_normal((n, T))).astype(np.float32), tq))
        vit_inputs.append((np.zeros((n, T), np.float32), np.where(tq < -1e11, tq, 0).astype(np.float32)))
    # large magnitudes (fp32 absorption)
    vit_inputs.append(((rng.standard_normal((9, T)) * 1e6).astype(np.float32), trans_pert))
    vit_inputs.append(((rng.standard_normal((9, T)) * 1e-3 + 3e4).astype(np.float32), trans_pert))
    for feats, trans in vit_inputs:
        path, conf = run_vit(feats, trans)
        cases["c%d_feats" % ci] = feats
        cases["c%d_trans" % ci] = trans
        cases["c%d_path" % ci] = path
        cases["c%d_conf" % ci] = conf
        ci += 1
    cases["n_cases"] = np.int64(ci)
    cases["start"], cases["stop"], cases["x_idx"] = np.int64(start), np.int64(stop), np.int64(x_idx)
    np.savez_compressed(os.path.join(GOLD, "viterbi.npz"), **cases)

    # ---------------- G5: _obtain_labels (remove_x re-padding) ----------------
    cases = {}
    B, n = 3, 14
    feats = (rng.standard_normal((B, n, T)) * 1.5).astype(np.float32)
    lengths = np.asarray([14, 10, 12], np.int64)
    nreal = (4, 10, 1)
    tags = np.zeros((B, n), np.int64)
    valid = [i for i in range(T) if i not in (start, stop, x_idx, 0)]
    for b in range(B):
        tags[b, :nreal[b]] = rng.choice(valid, size=nreal[b])
        tags[b, nreal[b]:lengths[b]] = x_idx
    with torch.no_grad():
        tagger.transitions.copy_(torch.from_numpy(trans_pert))
        sents = [_Sent(int(lengths[b]), tags[b]) for b in range(B)]
        mask = (torch.arange(n)[None, :] < torch.from_numpy(lengths)[:, None]).float()
        ft = torch.from_numpy(feats)
        # (a) evaluate() order: _calculate_loss first narrows self.mask (:2621-2622)
        tagger.mask = mask
        tagger._calculate_loss(ft, sents, mask)
        lab_a, _ = tagger._obtain_labels(ft, sents)
        # (b) speed_test order: self.mask is the plain length mask
        tagger.mask = mask
        lab_b, _ = tagger._obtain_labels(ft, sents)
    cases["feats"], cases["lengths"], cases["tags"], cases["trans"] = feats, lengths, tags, trans_pert
    for b in range(B):
        cases["a%d_tags" % b] = np.asarray([td.get_idx_for_item(l.value) for l in lab_a[b]], np.int32)
        cases["a%d_conf" % b] = np.asarray([l.score for l in lab_a[b]], np.float32)
        cases["b%d_tags" % b] = np.asarray([td.get_idx_for_item(l.value) for l in lab_b[b]], np.int32)
        cases["b%d_conf" % b] = np.asarray([l.score for l in lab_b[b]], np.float32)
    cases["start"], cases["stop"], cases["x_idx"] = np.int64(start), np.int64(stop), np.int64(x_idx)
    cases["items"] = np.asarray(items)
    np.savez_compressed(os.path.join(GOLD, "obtain_labels.npz"), **cases)

    # ---------------- G6: encoder (transformers 5.15 XLMRobertaModel, eager fp32) ----------------
    from transformers import XLMRobertaConfig, XLMRobertaModel
    from oracle import encoder as oenc

    cases = {}
    for tag, (V, H, L, A, F_, S, B) in {"tiny": (120, 64, 2, 4, 128, 24, 3), "wide": (64, 128, 1, 2, 256, 40, 2)}.items():
        hcfg = XLMRobertaConfig(vocab_size=V, hidden_size=H, num_hidden_layers=L, num_attention_heads=A,
                                intermediate_size=F_, max_position_embeddings=S + 2 + 8, type_vocab_size=1,
                                pad_token_id=1, layer_norm_eps=1e-5, hidden_act="gelu",
                                hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
        hcfg._attn_implementation = "eager"
        torch.manual_seed(7)
        model = XLMRobertaModel(hcfg, add_pooling_layer=False).eval()
        with torch.no_grad():
            for name, p in model.named_parameters():
                if p.dim() == 1:  # make biases / LN params non-trivial
                    p.add_(torch.randn_like(p) * 0.1)
                else:
                    p.mul_(3.0)
        ids = torch.from_numpy(rng.integers(3, V, size=(B, S))).long()
        ids[:, 0] = 0
        am = torch.ones(B, S, dtype=torch.long)
        # ragged: reference pads ids with 0 and mask with 0 (embeddings.py:3247-3260)
        for b in range(1, B):
            cut = S - 3 * b
            ids[b, cut - 1] = 2
            ids[b, cut:] = 0
            am[b, cut:] = 0
        ids[0, -1] = 2
        with torch.no_grad():
            out = model(input_ids=ids, attention_mask=am, output_hidden_states=True, return_dict=True)
        sd = {k: v.detach().numpy() for k, v in model.state_dict().items() if "position_ids" not in k and "token_type_ids" not in k}
        cases[tag + "_cfg"] = np.asarray([V, H, L, A, F_, S + 2 + 8], np.int64)
        cases[tag + "_ids"] = ids.numpy()
        cases[tag + "_mask"] = am.numpy()
        cases[tag + "_last"] = out.last_hidden_state.numpy()
        cases[tag + "_hs1"] = out.hidden_states[1].numpy()
        for k, v in sd.items():
            cases[tag + "/" + k] = v
        # sanity: the oracle restatement agrees with HF here
        ocfg = oenc.EncoderConfig(vocab_size=V, hidden_size=H, num_hidden_layers=L, num_attention_heads=A,
                                  intermediate_size=F_, max_position_embeddings=S + 2 + 8)
        params = {k: torch.from_numpy(v) for k, v in sd.items()}
        mine = oenc.encoder_forward(params, ocfg, ids, am)
        valid = am.bool()
        err = (mine - out.last_hidden_state)[valid].abs().max().item()
        print("encoder", tag, "oracle-vs-HF max abs err on unmasked positions:", err)
        assert err < 2e-5
    np.savez_compressed(os.path.join(GOLD, "encoder_tiny.npz"), **cases)

    # ---------------- G8/G9: optimiser grouping, AdamW, schedule ----------------
    # Built the way finetune_trainer.py:552-571,686-688 builds it, on a toy module whose parameter
    # NAMES match the tagger's (the grouping is by name).
    import transformers
    from transformers import get_linear_schedule_with_warmup

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.embeddings = torch.nn.Linear(6, 5)
            self.linear = torch.nn.Linear(5, 4)
            self.transitions = torch.nn.Parameter(torch.randn(4, 4))

    torch.manual_seed(99)
    toy = Toy()
    lr, lr_rate, t_total = 5e-6, 10000, 7
    finetune_params = {name: param for name, param in toy.named_parameters() if 'embedding' in name or name == 'linear.weight' or name == 'linear.bias'}
    other_params = {name: param for name, param in toy.named_parameters() if 'embedding' not in name and name != 'linear.weight' and name != 'linear.bias'}
    opt = transformers.AdamW([{"params": other_params.values(), "lr": lr * lr_rate},
                              {"params": finetune_params.values()}], lr=lr)
    sch = get_linear_schedule_with_warmup(opt, num_warmup_steps=0, num_training_steps=t_total)
    cases = {"lr": np.float64(lr), "lr_rate": np.float64(lr_rate), "t_total": np.int64(t_total)}
    names = [n for n, _ in toy.named_parameters()]
    cases["names"] = np.asarray(names)
    for n, p in toy.named_parameters():
        cases["p0/" + n] = p.detach().clone().numpy()
    lrs = []
    for step in range(3):
        for n, p in toy.named_parameters():
            g = torch.from_numpy(rng.standard_normal(tuple(p.shape)).astype(np.float32))
            p.grad = g
            cases["g%d/%s" % (step, n)] = g.numpy().copy()
        norm = torch.nn.utils.clip_grad_norm_(toy.parameters(), 5.0)
        cases["norm%d" % step] = np.float64(norm)
        lrs.append([g["lr"] for g in opt.param_groups])
        opt.step()
        sch.step()
        for n, p in toy.named_parameters():
            cases["p%d/%s" % (step + 1, n)] = p.detach().clone().numpy()
    cases["lrs"] = np.asarray(lrs, np.float64)
    np.savez_compressed(os.path.join(GOLD, "adamw.npz"), **cases)
    # ---------------- G11: reconstruct_tokens_from_subtokens (embeddings.py:3347) with the tiny test tokenizer ----------
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tiny_assets
    from flair.embeddings import TransformerWordEmbeddings as RefTWE
    from flair.data import Sentence as RefSentence
    tdir = tempfile.mkdtemp()
    tok = tiny_assets.build_tokenizer_dir(tdir)

    class _Stub:
        tokenizer = tok
        special_tokens = ["<s>"]
        _remove_special_markup = staticmethod(RefTWE._remove_special_markup)
        _get_processed_token_text = RefTWE._get_processed_token_text

    stub = _Stub()
    texts = ["alice visited berlin </s> the museum of art", "zalando research is located in berlin .",
             "Bob , Carol and ALICE ( 1999 ) -- new york city !", "a", "google founded by john smith 2021 </s> wikipedia article about history",
             "river thames near london </s> london capital population language people"]
    recs = []
    for t in texts:
        sent = RefSentence(t)
        pieces = tok.tokenize(sent.to_tokenized_string())
        lens = RefTWE.reconstruct_tokens_from_subtokens(stub, sent, pieces)
        recs.append({"text": t, "tokens": [x.text for x in sent], "pieces": pieces, "lengths": [int(x) for x in lens]})
    import json
    with open(os.path.join(GOLD, "subtoken_lengths.json"), "w") as f:
        json.dump(recs, f, indent=1, ensure_ascii=False)

    # ---------------- G10: get_spans + Metric on predicted vs gold tag sequences (data.py:455, training_utils.py:26) ------
    from flair.data import Sentence as RS
    from flair.training_utils import Metric as RefMetric
    tagsets = ["O", "B-LOC", "I-LOC", "E-LOC", "S-LOC", "B-PER", "E-PER", "S-PER", "S-X", "I-PER", "S-CORP"]
    recs = []
    metric = RefMetric("g10")
    for k in range(12):
        n = int(rng.integers(3, 12))
        words = ["w%d" % i for i in range(n)]
        gold = [str(x) for x in rng.choice(tagsets, size=n)]
        pred = [g if rng.random() < 0.6 else str(rng.choice(tagsets)) for g in gold]
        conf = [float(x) for x in np.round(rng.uniform(0.3, 1.0, size=n), 3)]
        s = RS(" ".join(words))
        for tkn, g_, p_, c_ in zip(s, gold, pred, conf):
            tkn.add_tag("ner", g_)
            tkn.add_tag("predicted", p_, c_)
        gs = [(sp.tag, str(sp), float(sp.score)) for sp in s.get_spans("ner")]
        ps = [(sp.tag, str(sp), float(sp.score)) for sp in s.get_spans("predicted")]
        for tag, text, _ in ps:
            if (tag, text) in [(a, b) for a, b, _ in gs]:
                metric.add_tp(tag)
            else:
                metric.add_fp(tag)
        for tag, text, _ in gs:
            if (tag, text) not in [(a, b) for a, b, _ in ps]:
                metric.add_fn(tag)
        recs.append({"words": words, "gold": gold, "pred": pred, "conf": conf, "gold_spans": gs, "pred_spans": ps})
    summary = {"classes": metric.get_classes(), "micro_f": metric.micro_avg_f_score(), "macro_f": metric.macro_avg_f_score(),
               "precision": metric.precision(), "recall": metric.recall(), "accuracy": metric.accuracy(),
               "per_class": {c: [metric.get_tp(c), metric.get_fp(c), metric.get_fn(c), metric.f_score(c)] for c in metric.get_classes()}}
    with open(os.path.join(GOLD, "spans_metric.json"), "w") as f:
        json.dump({"sentences": recs, "metric": summary}, f, indent=1)

    print("wrote fixtures to", GOLD)
    for f in sorted(os.listdir(GOLD)):
        print("  %-28s %8d bytes" % (f, os.path.getsize(os.path.join(GOLD, f))))


if __name__ == "__main__":
    main()
