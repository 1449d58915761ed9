"""CPU: the drop-in flair surface's host logic -- spans, Metric, Dictionary, tag schemes, CoNLL reader, batching, sub-token
alignment -- against known answers the reference's own tests pin (tests/test_data.py:468-573 test_spans, :179-251 Dictionary,
tests/test_utils.py:41-94 Metric) and golden vectors captured from the reference (tests/golden/*.json)."""
import json
import os

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
    ids, first = emb.tokenize_sentence(s)
    assert ids[0] == 0 and ids[-1] == 2                      # <s> ... </s>
    assert ids.count(2) == 2                                 # the <EOS> word token became the tokenizer's eos id
    assert first[0] == 1 and all(b > a for a, b in zip(first, first[1:]))
    ids_np, am, fi, lengths = emb.prepare_batch([s, Sentence("bob")])
    assert ids_np.shape == am.shape and am[1].sum() < am[0].sum() and ids_np[1, am[1].sum():].sum() == 0   # padded with 0
    assert fi.shape == (2, 6) and list(lengths) == [6, 1] and (fi[1, 1:] == -1).all()


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
