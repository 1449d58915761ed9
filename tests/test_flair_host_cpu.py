"""CPU: the drop-in flair surface's host logic -- spans, Metric, Dictionary, tag schemes, CoNLL reader, batching, sub-token
alignment -- against known answers the reference's own tests pin (tests/test_data.py:468-573 test_spans, :179-251 Dictionary,
tests/test_utils.py:41-94 Metric) and golden vectors captured from the reference (tests/golden/*.json)."""
import json
import os

import numpy as np
import pytest

from flair.data import Dictionary, Label, Sentence, iob2, iob_iobes
from flair.training_utils import Metric

SENT = "Zalando Research is located in Berlin ."


def _tagged(tags, scores=None):
    s = Sentence(SENT)
    for i, t in tags.items():
        s[i].add_tag("ner", t, 1.0 if scores is None else scores[i])
    return s


@pytest.mark.parametrize("tags,expected", [
    ({0: "B-ORG", 1: "E-ORG", 5: "S-LOC"}, [("Zalando Research", "ORG"), ("Berlin", "LOC")]),          # BIOES
    ({0: "B-ORG", 1: "I-ORG", 5: "B-LOC"}, [("Zalando Research", "ORG"), ("Berlin", "LOC")]),          # BIO
    ({0: "I-ORG", 1: "E-ORG", 5: "I-LOC"}, [("Zalando Research", "ORG"), ("Berlin", "LOC")]),          # broken openings
])
def test_spans_schemes(tags, expected):
    spans = _tagged(tags).get_spans("ner")
    assert [(s.text, s.tag) for s in spans] == expected


def test_spans_untyped_and_mixed_tags():
    s = _tagged({0: "I-ORG", 1: "E-ORG", 2: "aux", 3: "verb", 4: "preposition", 5: "I-LOC"})
    sp = s.get_spans("ner")
    assert len(sp) == 5 and (sp[0].text, sp[0].tag) == ("Zalando Research", "ORG") and (sp[4].text, sp[4].tag) == ("Berlin", "LOC")
    s = _tagged({0: "I-ORG", 1: "S-LOC", 2: "aux", 3: "B-relation", 4: "E-preposition", 5: "S-LOC"})
    sp = s.get_spans("ner")
    assert len(sp) == 5
    assert (sp[0].text, sp[0].tag) == ("Zalando", "ORG") and (sp[1].text, sp[1].tag) == ("Research", "LOC")
    assert (sp[3].text, sp[3].tag) == ("located in", "relation")


def test_spans_adjacent_single_then_multi():
    s = Sentence("after three Irish Republican Army mortar bombs")
    s[2].add_tag("ner", "S-MISC")
    s[3].add_tag("ner", "B-MISC")
    s[4].add_tag("ner", "E-MISC")
    assert [x.text for x in s.get_spans("ner")] == ["Irish", "Republican Army"]


def test_span_scores_and_threshold():
    s = _tagged({0: "B-ORG", 1: "E-ORG", 5: "S-LOC"}, scores={0: 1.0, 1: 0.9, 5: 0.5})
    sp = s.get_spans("ner", min_score=0.0)
    assert [(x.text, x.tag, x.score) for x in sp] == [("Zalando Research", "ORG", 0.95), ("Berlin", "LOC", 0.5)]
    assert len(s.get_spans("ner", min_score=0.6)) == 1
    assert len(s.get_spans("ner", min_score=0.99)) == 0


def test_metric_rounding_semantics():
    m = Metric("Test")
    for c in ("class-1", "class-2", "class-4"):
        m.add_tp(c); m.add_tn(c); m.add_tn(c); m.add_fp(c)
    for _ in range(10):
        m.add_tp("class-3")
    for _ in range(90):
        m.add_fp("class-3")
    assert [m.precision(c) for c in ("class-1", "class-2", "class-3", "class-4")] == [0.5, 0.5, 0.1, 0.5]
    assert all(m.recall(c) == 1 for c in m.get_classes())
    assert [m.f_score(c) for c in ("class-1", "class-2", "class-3", "class-4")] == [0.6667, 0.6667, 0.1818, 0.6667]
    assert [m.accuracy(c) for c in ("class-1", "class-2", "class-3", "class-4")] == [0.5, 0.5, 0.1, 0.5]
    assert m.micro_avg_f_score() == 0.2184 == m.f_score()
    assert m.macro_avg_f_score() == 0.5454749999999999
    assert m.micro_avg_accuracy() == 0.1226 == m.accuracy() and m.macro_avg_accuracy() == 0.4
    assert m.precision() == 0.1226 and m.recall() == 1


def test_dictionary_behaviour(tmp_path):
    d = Dictionary()
    assert d.get_idx_for_item("<unk>") == 0 and len(d) == 1
    assert d.add_item("class_1") == 1 and d.add_item("class_2") == 2 and d.add_item("class_1") == 1
    assert d.get_idx_for_item("nope") == 0
    assert d.get_items() == ["<unk>", "class_1", "class_2"] and d.get_item_for_index(2) == "class_2"
    d2 = Dictionary(add_unk=False)
    d2.add_item("a")
    assert d2.get_idx_for_item("a") == 0
    f = tmp_path / "dict.pkl"
    d.save(f)
    assert Dictionary.load_from_file(str(f)).item2idx == d.item2idx


def test_tag_scheme_conversion():
    tags = [Label(t) for t in ["I-PER", "I-PER", "O", "B-LOC", "I-LOC", "I-ORG", "B-X", "B-X"]]
    assert iob2(tags)
    assert [t.value for t in tags] == ["B-PER", "I-PER", "O", "B-LOC", "I-LOC", "B-ORG", "B-X", "B-X"]
    assert iob_iobes(tags) == ["B-PER", "E-PER", "O", "B-LOC", "E-LOC", "S-ORG", "S-X", "S-X"]


def test_spans_and_metric_vs_reference_golden(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "spans_metric.json")))
    m = Metric("g10")
    for rec in g["sentences"]:
        s = Sentence(" ".join(rec["words"]))
        for tok, gt, pt, c in zip(s, rec["gold"], rec["pred"], rec["conf"]):
            tok.add_tag("ner", gt)
            tok.add_tag("predicted", pt, c)
        gs = [(x.tag, str(x), x.score) for x in s.get_spans("ner")]
        ps = [(x.tag, str(x), x.score) for x in s.get_spans("predicted")]
        assert [(a, b) for a, b, _ in gs] == [(a, b) for a, b, _ in rec["gold_spans"]]
        assert [(a, b) for a, b, _ in ps] == [(a, b) for a, b, _ in rec["pred_spans"]]
        assert [c for _, _, c in ps] == pytest.approx([c for _, _, c in rec["pred_spans"]])
        gk, pk = [(a, b) for a, b, _ in gs], [(a, b) for a, b, _ in ps]
        for k in pk:
            (m.add_tp if k in gk else m.add_fp)(k[0])
        for k in gk:
            if k not in pk:
                m.add_fn(k[0])
    r = g["metric"]
    assert m.get_classes() == r["classes"]
    assert (m.micro_avg_f_score(), m.macro_avg_f_score(), m.precision(), m.recall(), m.accuracy()) == \
        (r["micro_f"], r["macro_f"], r["precision"], r["recall"], r["accuracy"])
    for c, (tp, fp, fn, f) in r["per_class"].items():
        assert (m.get_tp(c), m.get_fp(c), m.get_fn(c), m.f_score(c)) == (tp, fp, fn, f)


def test_subtoken_alignment_vs_reference_golden(golden_dir, tmp_path):
    import tiny_assets
    from flair.embeddings import TransformerWordEmbeddings
    d = tiny_assets.build_model_dir(str(tmp_path / "m"))
    emb = TransformerWordEmbeddings(model=d, layers="-1", pooling_operation="first", fine_tune=True)
    for rec in json.load(open(os.path.join(golden_dir, "subtoken_lengths.json"))):
        sent = Sentence(rec["text"])
        pieces = emb.tokenizer.tokenize(sent.to_tokenized_string())
        assert pieces == rec["pieces"], "test tokenizer drifted from the one the fixture was generated with"
        assert emb.reconstruct_tokens_from_subtokens(list(sent), pieces) == rec["lengths"]


def test_eos_token_and_first_subtoken_index(tmp_path):
    import tiny_assets
    from flair.embeddings import TransformerWordEmbeddings
    d = tiny_assets.build_model_dir(str(tmp_path / "m"))
    emb = TransformerWordEmbeddings(model=d, layers="-1", pooling_operation="first", fine_tune=True)
    s = Sentence("alice visited berlin <EOS> the museum")
    rows, first_row, first = emb.tokenize_sentence(s)
    assert len(rows) == 1 and set(first_row) == {0}          # fits one window
    ids = rows[0]
    assert ids[0] == 0 and ids[-1] == 2                      # <s> ... </s>
    assert ids.count(2) == 2                                 # the <EOS> word token became the tokenizer's eos id
    assert first[0] == 1 and all(b > a for a, b in zip(first, first[1:]))
    ids_np, am, fi, lengths, frow = emb.prepare_batch([s, Sentence("bob")])
    assert ids_np.shape == am.shape and am[1].sum() < am[0].sum() and ids_np[1, am[1].sum():].sum() == 0   # padded with 0
    assert fi.shape == (2, 6) and list(lengths) == [6, 1] and (fi[1, 1:] == -1).all()
    assert (frow[0] == 0).all() and frow[1, 0] == 1


@pytest.mark.parametrize("n_words", [40, 75, 140, 260])
def test_sliding_window_matches_reference_stitching(tmp_path, n_words):
    """Sentences longer than one window: the (row, position) our batch gathers each word token's first sub-token from must
    be the element the reference reaches by concatenating window states with its seam rule
    (flair/embeddings.py:3292-3299: acc[:-1-stride//2] ++ next[1+stride//2:]) and walking sub-token counts from offset 1."""
    import tiny_assets
    from flair.embeddings import TransformerWordEmbeddings
    d = tiny_assets.build_model_dir(str(tmp_path / "m"))
    emb = TransformerWordEmbeddings(model=d, layers="-1", pooling_operation="first", fine_tune=True)
    emb.max_subtokens_sequence_length, emb.stride = 64, 32          # small windows keep the test fast; same arithmetic
    rng = np.random.default_rng(n_words)
    words = [str(w) for w in rng.choice(tiny_assets.WORDS, size=n_words)]
    s = Sentence(" ".join(words))
    rows, first_row, first = emb.tokenize_sentence(s)
    pieces = emb.tokenizer.tokenize(" ".join(words))
    counts = emb.reconstruct_tokens_from_subtokens(list(s), pieces)
    content = emb.tokenizer.convert_tokens_to_ids(pieces)
    W, st = 62, 32
    # the windows encode_plus(max_length=64, stride=32, return_overflowing_tokens=True) yields, re-fed with its overflow
    exp_rows, rest = [], list(content)
    while rest:
        exp_rows.append([0] + rest[:W] + [2])
        rest = rest[W - st:] if len(rest) > W else None
    assert rows == exp_rows
    assert len(rows) > 1 or len(content) <= W
    # reference stitching on "hidden states" that name their own (row, position)
    acc = [(0, p) for p in range(len(rows[0]))]
    for r in range(1, len(rows)):
        acc = acc[:-1 - st // 2] + [(r, p) for p in range(len(rows[r]))][1 + st // 2:]
    assert len(acc) == len(content) + 2                              # the seams neither drop nor duplicate a sub-token
    start = 1
    for k, c in enumerate(counts):
        if c == 0:
            assert first[k] == -1
            continue
        assert (first_row[k], first[k]) == acc[start], (k, first_row[k], first[k], acc[start])
        assert rows[first_row[k]][first[k]] == content[start - 1]     # and that element really is the token's first piece
        start += c
    ids_np, am, fi, lengths, frow = emb.prepare_batch([Sentence("bob"), s])
    assert ids_np.shape[0] == 1 + len(rows) and (frow[1, :len(s)] >= 1).all() and frow[0, 0] == 0


def test_column_corpus_and_loader(tmp_path):
    import tiny_assets
    from flair.custom_data_loader import BatchedData, ColumnDataLoader
    from flair.datasets import ColumnCorpus
    folder = tiny_assets.write_conll_corpus(str(tmp_path / "c"))
    corpus = ColumnCorpus(folder, {0: "text", 1: "pos", 2: "upos", 3: "ner"}, tag_to_bioes="ner", comment_symbol="# id")
    assert (len(corpus.train), len(corpus.dev), len(corpus.test)) == (24, 8, 8)
    s = corpus.train[0]
    texts = [t.text for t in s]
    k = texts.index("<EOS>")
    assert all(t.get_tag("ner").value == "S-X" for t in s.tokens[k:])      # B-X context -> S-X after IOBES
    assert all(not t.get_tag("ner").value.endswith("-X") for t in s.tokens[:k])
    assert not any(t.startswith("#") for t in texts)                      # comment lines skipped
    td = corpus.make_tag_dictionary("ner")
    assert td.get_items()[:2] == ["<unk>", "O"] and td.get_items()[-2:] == ["<START>", "<STOP>"]
    loader = ColumnDataLoader(list(corpus.train), 5, sentence_level_batch=True)
    assert sum(len(b) for b in loader) == 24 and all(len(b) <= 5 for b in loader)
    lens = [len(x) for b in loader for x in b]
    assert lens == sorted(lens)                                           # ascending word-token length
    loader.assign_tags("ner", td)
    b0 = loader[0]
    assert isinstance(b0, BatchedData) and b0.ner_tags.shape == (len(b0), max(len(x) for x in b0))
    order = [id(b) for b in loader.data]
    loader.reshuffle()
    assert sorted(order) == sorted(id(b) for b in loader.data)            # batch membership fixed, only order changes


def test_context_file_format_writer_validator_and_reader(tmp_path):
    """write a knowledge-augmented CoNLL file with the reference's convention and budget rule, validate it,
    and read it back through ColumnCorpus (context B-X -> S-X, comment lines skipped)."""
    import tiny_assets
    from flair.datasets import ColumnCorpus
    from kbner import context_format as cf
    tok = tiny_assets.build_tokenizer_dir(str(tmp_path / "tok"))
    count = lambda text: len(tok.tokenize(text))  # noqa: E731
    rng = np.random.default_rng(0)
    sents = []
    for i in range(12):
        words = [str(w) for w in rng.choice(tiny_assets.WORDS, size=int(rng.integers(3, 9)))]
        ner = ["O"] * len(words)
        ner[0] = "B-LOC"
        ctx = [" ".join(str(w) for w in rng.choice(tiny_assets.WORDS, size=int(rng.integers(5, 60)))) for _ in range(int(rng.integers(0, 6)))]
        sents.append(dict(id="s%d" % i, tokens=[(w, "_", "_", t) for w, t in zip(words, ner)], contexts=ctx))
    folder = tmp_path / "c"
    folder.mkdir()
    for name in ("train.txt", "dev.txt", "test.txt"):
        cf.write_file(str(folder / name), sents, count, length_limit=64)
    st = cf.validate_file(str(folder / "train.txt"), count, length_limit=64)
    assert st["sentences"] == 12 and st["over_budget"] == 0 and 0 < st["max_subtokens"] <= 64
    assert 0 < st["with_context"] < 12                      # sentences without retrievals carry no <EOS>
    # budget rule: a context that does not fit is skipped, a later shorter one may still be taken
    used = cf.select_contexts(10, ["a " * 30, "b c", "d " * 40, "e"], lambda t: len(t.split()), length_limit=24)
    assert used == ["b c", "e"]
    corpus = ColumnCorpus(str(folder), {0: "text", 1: "pos", 2: "upos", 3: "ner"}, tag_to_bioes="ner", comment_symbol="# id")
    assert len(corpus.train) == 12
    for s in corpus.train:
        texts = [t.text for t in s]
        if "<EOS>" in texts:
            k = texts.index("<EOS>")
            assert all(t.get_tag("ner").value == "S-X" for t in s.tokens[k:])
    bad = folder / "bad.txt"
    bad.write_text("berlin _ _ B-LOC\n<EOS> B-X B-X B-X\nfoo _ _ O\n\n")
    with pytest.raises(cf.FormatError):
        cf.validate_file(str(bad))


def test_get_spans_skip_class_equals_filtering():
    """evaluate()'s remove_x shortcut: spans extracted with skip_class='X' == all spans with the X ones filtered out, for
    gold-like rows (S-X context tail) and arbitrary predicted rows"""
    rng = np.random.default_rng(4)
    tags = ["O", "B-LOC", "I-LOC", "E-LOC", "S-LOC", "B-PER", "E-PER", "S-PER", "S-X", "I-PER"]
    for _ in range(200):
        n = int(rng.integers(1, 14))
        row = [str(rng.choice(tags)) for _ in range(n)]
        if rng.random() < 0.7:
            row += ["S-X"] * int(rng.integers(1, 6))
        s = Sentence(" ".join("w%d" % i for i in range(len(row))))
        for tok, t in zip(s, row):
            tok.add_tag("ner", t)
        full = [(sp.tag, str(sp)) for sp in s.get_spans("ner") if sp.tag != "X"]
        fast = [(sp.tag, str(sp)) for sp in s.get_spans("ner", skip_class="X") if sp.tag != "X"]
        assert full == fast, (row, full, fast)


def test_multi_view_pairing_and_weights_host_logic():
    """ModelFinetuner's multi-view bookkeeping without a GPU: the *doc* corpus is paired with the source corpus sentence by
    sentence (finetune_trainer.py:316-344) and an accumulation group's per-sentence weights put (1 - rate) on the NLL of the
    micro-batches that carry a second view, rate / |paired sentences of that micro-batch| on their KL terms (:909-914,959-966)"""
    from flair.trainers.finetune_trainer import ModelFinetuner

    class S:
        def __init__(self, name):
            self.name = name

    class Corpus:
        targets = ["ColumnCorpus-NEWS", "ColumnCorpus-NEWSDOC"]
        train_list = [[S("a0"), S("a1"), S("a2")], [S("d0"), S("d1"), S("d2")]]
        dev_list = [[S("b0")], [S("e0")]]
        test_list = [[S("c0")], [S("f0")]]

    class Model:
        multi_view_training = True

        def multi_view_plan(self, batch, with_flag=False):   # paired sentences that have context (+ the batch-level rule, one walk)
            sel = [i for i, s in enumerate(batch) if hasattr(s, "orig_sent") and not getattr(s, "no_x", False)]
            if self.check_multi_view(batch) is False:
                sel = []
            return (self.check_multi_view(batch) is not False, sel) if with_flag else sel

        def check_multi_view(self, batch):  # reference rule: some sentence is paired AND some S-X tag occurs in the batch
            if not any(hasattr(s, "orig_sent") for s in batch):
                return False
            return True if any(not getattr(s, "no_x", False) for s in batch) else False

    t = ModelFinetuner.__new__(ModelFinetuner)
    t.model, t.corpus = Model(), Corpus()
    t.corpus2id = {n: i for i, n in enumerate(Corpus.targets)}
    t._pair_multi_view_corpora()
    assert [s.orig_sent.name for s in Corpus.train_list[1]] == ["a0", "a1", "a2"]
    assert Corpus.dev_list[1][0].orig_sent.name == "b0" and Corpus.test_list[1][0].orig_sent.name == "c0"
    assert not any(hasattr(s, "orig_sent") for s in Corpus.train_list[0])
    a, d = Corpus.train_list
    group = [[a[0], a[1]], [d[0], a[2]], [d[1], d[2]]]            # plain, mixed, all paired
    wts, mv = t._group_weights(group, 0.25)
    G = 3
    assert np.allclose(wts, [1 / (G * 2)] * 2 + [0.75 / (G * 2)] * 2 + [0.75 / (G * 2)] * 2)
    assert mv[0] == [2, 4, 5] and np.allclose(mv[1], [0.25 / (G * 1), 0.25 / (G * 2), 0.25 / (G * 2)])
    wts, mv = t._group_weights(group[:1], 0.25)
    assert mv is None and np.allclose(wts, [0.5, 0.5])
    wts, mv = t._group_weights(group, None)                        # multi-view off: the plain fused weights
    assert mv is None and np.allclose(wts, [1 / 6] * 6)
    # a paired sentence WITHOUT context tags next to an unpaired one WITH them: check_multi_view is a tensor in the reference
    # (finetune_trainer.py:909-914), so the NLL is scaled by (1 - rate) although no sentence enters the KL term
    d[0].no_x = True
    wts, mv = t._group_weights([[d[0], a[2]]], 0.25)
    assert mv is None and np.allclose(wts, [0.75 / 2] * 2)


def test_batch_spans_equals_get_spans_on_random_tag_sequences():
    """flair.data.batch_spans (numpy over tag-id arrays, what evaluate() uses) == Sentence.get_spans (the reference's per-token
    rule, flair/data.py:455-532) on random tag sequences: malformed BIOES (I- after O, E- without B-, S- followed by I- of the
    same / another class), long runs of single-token S-X context, bare class names, both remove_x filters, min_score."""
    from flair.data import Label, Sentence, batch_spans, span_tables
    rng = np.random.default_rng(7)
    items = ["<unk>", "O", "B-PER", "I-PER", "E-PER", "S-PER", "B-LOC", "I-LOC", "E-LOC", "S-LOC", "S-X", "B-X", "MISC", "<START>", "<STOP>"]
    tbl = span_tables(items)
    for trial in range(60):
        B = int(rng.integers(1, 6))
        n = int(rng.integers(1, 40))
        lens = rng.integers(1, n + 1, size=B)
        lens[0] = n
        p = np.ones(len(items))
        p[1] = 4.0
        p[10] = 6.0 if trial % 2 else 1.0          # S-X-heavy in half of the trials
        ids = rng.choice(len(items), size=(B, n), p=p / p.sum())
        scores = np.round(rng.uniform(0.2, 1.0, size=(B, n)), 3)
        sents = []
        for b in range(B):
            s = Sentence(" ".join("w%d" % i for i in range(int(lens[b]))))
            for i, tok in enumerate(s.tokens):
                tok.add_tag_label("t", Label(items[ids[b, i]], float(scores[b, i])))
            sents.append(s)
        drop = rng.random((B, n)) < 0.15
        for kw_fast, kw_ref in ((dict(), dict()), (dict(skip_class="X"), dict(skip_class="X")), (dict(min_score=0.6), dict(min_score=0.6)),
                                (dict(drop_flags=drop), None)):
            got = batch_spans(sents, ids, scores, tbl, **kw_fast)
            for b, s in enumerate(sents):
                if kw_ref is None:
                    ref = s.get_spans("t", drop_touching=set((np.nonzero(drop[b, :len(s)])[0] + 1).tolist()))
                else:
                    ref = s.get_spans("t", **kw_ref)
                assert [(x.tag, str(x), x.score) for x in got[b]] == [(x.tag, str(x), x.score) for x in ref], (trial, b, kw_fast)


def test_kd_batch_assembly_reproduces_the_reference_loss():
    """FastSequenceTagger._kd_batch -- per-sentence teacher targets (host arrays trimmed to the sentence) -> the padded batch
    tensors of the KD loss: concatenation over teachers, path weights, distill_with_gold reweighting (gold_const / exp_score).
    Checked end to end on the CPU: feeding its output to the fp32 oracle restatement (oracle/kd.py) must reproduce the loss the
    REFERENCE computed for the same sentences (tests/golden/kd_loss.npz, simple_forward_distillation_loss under autograd)."""
    import os
    import torch
    import tiny_assets
    from flair.models import FastSequenceTagger
    from oracle import kd as okd
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "kd_loss.npz"))
    start, stop, x_idx = int(g["start"]), int(g["stop"]), int(g["x_idx"])
    seen_gold = 0
    for c in range(int(g["n_cases"])):
        fake, sents, hb = tiny_assets.kd_golden_batch(g, c)
        kd = FastSequenceTagger._kd_batch(fake, sents, hb)
        B, n, T = g["c%d_es" % c].shape
        kw = {}
        if "scores" in kd:
            assert len(kd["scores"]) == int(g["c%d_n_teachers" % c]) and tuple(kd["scores"][0].shape) == (B, n, T)
            kw["scores_t"] = [x.cpu() for x in kd["scores"]]
        if "targets" in kd:
            assert tuple(kd["targets"].shape) == (B, n, int(g["c%d_best_k" % c]) * int(g["c%d_n_teachers" % c]))
            kw["targets"] = kd["targets"].cpu().long()
            if "weights" in kd:
                kw["weights"], kw["att_nums"] = kd["weights"].cpu(), kd["att_nums"]
                if fake.distill_with_gold:
                    seen_gold += 1
                    raw = torch.from_numpy(np.stack([s.get_teacher_weights() for s in sents], 0))
                    want = okd.gold_reweighted(raw, kw["targets"], g["c%d_tags" % c], g["c%d_lens" % c], fake.gold_const, fake.exp_score,
                                               kd["att_nums"])
                    assert float((kw["weights"] - want).abs().max()) < 1e-6
        if "exact" in kd:
            kw["exact"] = tuple(x.cpu() for x in kd["exact"])
        loss = okd.kd_loss(torch.from_numpy(g["c%d_es" % c]), torch.from_numpy(g["trans_s"]), g["c%d_lens" % c], g["c%d_tags" % c], start,
                           stop, x_idx, float(g["c%d_tau" % c]), float(g["c%d_interpolation" % c]), **kw)
        ref = float(g["c%d_loss" % c])
        assert abs(float(loss) - ref) <= 3e-5 * max(1.0, abs(ref)), (c, float(loss), ref)
    assert seen_gold == 2


def test_kd_batch_assembly_emission_reproduces_the_reference_loss():
    """the same for distill_emission (+ distill_prob, + distill_posterior): FastSequenceTagger._kd_batch's "emission" entry fed to
    the oracle reproduces the loss the REFERENCE computed (tests/golden/kd_emission.npz)"""
    import os
    import torch
    import tiny_assets
    from flair.models import FastSequenceTagger
    from oracle import kd as okd
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "kd_emission.npz"))
    start, stop, x_idx = int(g["start"]), int(g["stop"]), int(g["x_idx"])
    for c in range(int(g["n_cases"])):
        fake, sents, hb = tiny_assets.kd_emission_golden_batch(g, c)
        kd = FastSequenceTagger._kd_batch(fake, sents, hb)
        teach, is_prob = kd["emission"]
        assert tuple(teach.shape) == g["c%d_es" % c].shape and is_prob == bool(g["c%d_flags" % c][0])
        kw = {"emission": teach.cpu(), "emission_is_prob": is_prob}
        if "scores" in kd:
            kw["scores_t"] = [x.cpu() for x in kd["scores"]]
        loss = okd.kd_loss(torch.from_numpy(g["c%d_es" % c]), torch.from_numpy(g["trans_s"]), g["c%d_lens" % c], g["c%d_tags" % c], start,
                           stop, x_idx, float(g["c%d_tau" % c]), float(g["c%d_interpolation" % c]), **kw)
        ref = float(g["c%d_loss" % c])
        assert abs(float(loss) - ref) <= 3e-5 * max(1.0, abs(ref)), (c, float(loss), ref)


def test_distill_emission_switch_combinations():
    """the emission-level KD switches are accepted for a CRF student where the reference's loss is defined, and refused with the
    reason where it is not (checked before any device work): next to distill_crf / distill_exact the reference's trainer stores
    n-best / pairwise targets only and its loss then stacks empty prediction lists (sequence_tagger_model.py:2358); the multi-view
    loss has no emission term"""
    from flair.data import Dictionary
    from flair.models import FastSequenceTagger
    from flair.trainers import ModelFinetuner
    td = Dictionary(add_unk=True)
    for t in ("O", "B-X", "S-X", "<START>", "<STOP>"):
        td.add_item(t)
    base = dict(hidden_size=8, embeddings=None, tag_dictionary=td, tag_type="ner", use_crf=True, use_rnn=False)
    with pytest.raises(ValueError, match="needs distill_posterior"):
        FastSequenceTagger(distill_emission=True, distill_crf=True, **base)
    with pytest.raises(ValueError, match="needs distill_posterior"):
        FastSequenceTagger(distill_emission=True, distill_exact=True, **base)
    with pytest.raises(NotImplementedError, match="multi_view_training"):
        FastSequenceTagger(distill_emission=True, multi_view_training=True, distill_posterior=True, remove_x=True, **base)
    with pytest.raises(NotImplementedError, match="use_rnn: false"):
        FastSequenceTagger(distill_emission=True, **dict(base, use_rnn=True))
    # the trainer's gate: a CRF student in distill_mode needs one of the KD switches, distill_emission now among them
    import types
    student = types.SimpleNamespace(distill_crf=False, distill_exact=False, distill_posterior=False, distill_emission=False,
                                    multi_view_training=False)
    teacher = types.SimpleNamespace(eval=lambda: None)
    with pytest.raises(NotImplementedError, match="distill_emission"):
        ModelFinetuner(student, [teacher], corpus=None, distill_mode=True)
    student.distill_emission = True
    tr = ModelFinetuner(student, [teacher], corpus=None, distill_mode=True)
    assert tr.distill_mode and tr.teachers == [teacher]

