"""CPU: host-side pieces of the path against goldens captured by RUNNING THE REFERENCE's whole stack on the tiny KB-NER-shaped
corpus (oracle/gen_golden_e2e.py): CoNLL reader + BIOES + tag dictionary + ColumnDataLoader (a19/a20), the pooling index of
TransformerWordEmbeddings incl. dropped and clamped tokens (G7, a5), and FastSequenceTagger.evaluate's remove_x post-filter,
prediction lines and Result (G12, a15) with the decode stubbed by the reference's own predictions."""
import json
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


@pytest.fixture(scope="module")
def e2e():
    return json.load(open(os.path.join(GOLD, "e2e_train.json"), encoding="utf-8"))


@pytest.fixture(scope="module")
def workdir(tmp_path_factory, e2e):
    import tiny_assets
    d = tmp_path_factory.mktemp("e2e_cpu")
    cfg = tiny_assets.e2e_config(str(d), **e2e["config_kwargs"])
    return d, cfg


def _rec(s, tag="ner"):
    return {"tokens": [t.text for t in s], "tags": [t.get_tag(tag).value for t in s]}


def test_reader_dictionary_and_loader_match_the_reference(workdir):
    from flair.custom_data_loader import ColumnDataLoader
    from flair.datasets import ColumnCorpus
    from flair.list_data import ListCorpus
    d, cfg = workdir
    g = json.load(open(os.path.join(GOLD, "loader_reader.json")))
    cc = ColumnCorpus(**cfg["ner"]["ColumnCorpus-TINY"])
    for part, ds in (("train", cc.train), ("dev", cc.dev), ("test", cc.test)):
        assert [_rec(s) for s in ds] == g["corpus"][part], part          # tokens, `# id` comments skipped, B-X -> S-X (BIOES)
    corpus = ListCorpus(train=[cc.train], dev=[cc.dev], test=[cc.test], targets=["ColumnCorpus-TINY"])
    td = corpus.make_tag_dictionary(tag_type="ner")
    assert td.get_items() == g["tag_dictionary"]                           # same item ORDER = same tag indices
    train = list(cc.train)
    pos = {id(s): i for i, s in enumerate(train)}
    for bs in (1, 4):
        dl = ColumnDataLoader(train, bs, False, use_bert=False, sort_data=True, sentence_level_batch=True)
        dl.assign_tags("ner", td)
        ref = g["loaders"][str(bs)]
        assert [[pos[id(s)] for s in b] for b in dl] == ref["batches"]
        assert [b.ner_tags.tolist() for b in dl] == ref["ner_tags"]
        assert dl.num_examples == ref["num_examples"]
    dl = ColumnDataLoader(train, 4, False, use_bert=False, sort_data=False, sentence_level_batch=True)
    assert [[pos[id(s)] for s in b] for b in dl] == g["loaders"]["4_unsorted"]["batches"]
    dl = ColumnDataLoader(train, 40, False, use_bert=False, sort_data=True, sentence_level_batch=False)
    assert [[pos[id(s)] for s in b] for b in dl] == g["loaders"]["40_tokens"]["batches"]


@pytest.mark.parametrize("kind", ["sp", "wp"])
def test_pooling_index_matches_reference_features(workdir, kind, tmp_path):
    """G7: ids / mask the mirror feeds the encoder == the reference's, and gathering the reference's last hidden state with the
    mirror's (row, position) index reproduces the reference's features[B,n,H] -- `sp`: sentencepiece-style tokenizer, a deleted
    soft-hyphen token + a clamped long token; `wp`: BERT-style tokenizer whose dropped control-character tokens get NO sub-token
    and pool to zero vectors (embeddings.py:3306-3308), <EOS> -> [SEP]"""
    import tiny_assets
    from flair.data import Sentence
    from flair.embeddings import TransformerWordEmbeddings
    d, cfg = workdir
    z = np.load(os.path.join(GOLD, "pooling.npz"))
    g = {k[3:]: z[k] for k in z.files if k.startswith(kind + "/")}
    if kind == "sp":
        mdir = os.path.join(str(d), "xlmr-tiny")
    else:
        mdir = tiny_assets.build_model_dir(str(tmp_path / "bert-tiny"), tokenizer="wordpiece", seed=3)
    emb = TransformerWordEmbeddings(model=mdir, layers="-1", pooling_operation="first", fine_tune=True,
                                    maximum_subtoken_length=int(g["maximum_subtoken_length"]))
    sents = [Sentence(str(t)) for t in g["texts"]]
    assert [len(s) for s in sents] == g["n_tokens"].tolist()
    ids, am, first, lengths, first_row = emb.prepare_batch(sents)
    np.testing.assert_array_equal(ids, g["ids"])
    np.testing.assert_array_equal(am, g["mask"])
    hidden, feats = g["hidden"], g["features"]
    B, n, H = feats.shape
    mine = np.zeros_like(feats)
    for b in range(B):
        for i in range(int(lengths[b])):
            if first[b, i] >= 0:
                mine[b, i] = hidden[first_row[b, i], first[b, i]]
    np.testing.assert_array_equal(mine, feats)
    if kind == "wp":   # the crafted zero-vector cases really are in there
        zero = [(np.abs(feats[b, :int(lengths[b])]).sum(-1) == 0).sum() for b in range(B)]
        assert zero == [1, 2, 0], zero
        assert [(first[b, :int(lengths[b])] < 0).sum() for b in range(B)] == [1, 2, 0]


class _StubTagger:
    """FastSequenceTagger.evaluate with the three device calls replaced by the reference's captured predictions"""

    def __new__(cls, td, lines_by_sentence):
        from flair.data import Label
        from flair.models import FastSequenceTagger
        self = FastSequenceTagger.__new__(FastSequenceTagger)
        torch.nn.Module.__init__(self)
        self.tag_type, self.remove_x, self.tag_dictionary = "ner", True, td
        self.mask = None
        it = iter(lines_by_sentence)
        self.forward = lambda batch, prediction_mode=False: None
        self._calculate_loss = lambda feats, batch, mask: torch.tensor(0.0)

        def obtain(feats, batch, get_all_tags=False):
            out = []
            for s in batch:
                rows = next(it)
                assert [r[0] for r in rows] == [t.text for t in s]
                out.append([Label(r[2], float(r[3]) if "." in r[3] or "e" in r[3] else int(r[3])) for r in rows])
            return out, []

        self._obtain_labels = obtain
        return self


@pytest.mark.parametrize("part", ["dev", "test"])
def test_evaluate_lines_filter_and_result_match_the_reference(e2e, part, tmp_path):
    from flair.custom_data_loader import BatchedData
    from flair.data import Dictionary, Sentence
    g = e2e["evaluate"][part]
    gl = json.load(open(os.path.join(GOLD, "loader_reader.json")))
    td = Dictionary(add_unk=False)
    for it in gl["tag_dictionary"]:
        td.add_item(it)
    # the reference's lines, grouped per sentence
    per_sentence, cur = [], []
    for ln in g["lines"]:
        if ln == "":
            if cur:
                per_sentence.append(cur)
            cur = []
        else:
            cur.append(ln.split(" "))
    batches = []
    for b in g["batches"]:
        sents = []
        for r in b:
            s = Sentence(" ".join(r["tokens"]))
            for t, tg in zip(s, r["tags"]):
                t.add_tag("ner", tg)
            sents.append(s)
        batches.append(BatchedData(sents))
    assert sum(len(b) for b in batches) == len(per_sentence)
    tagger = _StubTagger(td, per_sentence)
    out = tmp_path / "pred.tsv"
    res, loss = tagger.evaluate(batches, out_path=out, embeddings_storage_mode="none")
    assert open(out, encoding="utf-8").read().split("\n") == g["lines"]
    assert res.log_line == g["log_line"] and res.log_header == g["log_header"]
    assert res.main_score == g["main_score"] and res.macro_score == g["macro_score"]
    assert res.detailed_results == g["detailed_results"]


def test_speed_test_order_writes_nothing_and_scores_nothing(e2e, tmp_path):
    """sequence_tagger_model.py:2618-2622: with speed_test the loss, the prediction lines and the metric are all skipped"""
    from flair.custom_data_loader import BatchedData
    from flair.data import Dictionary, Sentence
    g = e2e["evaluate"]["dev"]
    td = Dictionary(add_unk=False)
    per_sentence = [[[t, tg, "O", "1.0"] for t, tg in zip(r["tokens"], r["tags"])] for b in g["batches"] for r in b]
    batches = [BatchedData([Sentence(" ".join(r["tokens"])) for r in b]) for b in g["batches"]]
    tagger = _StubTagger(td, per_sentence)
    tagger._calculate_loss = None   # must not be called
    res, loss = tagger.evaluate(batches, out_path=tmp_path / "x.tsv", speed_test=True)
    assert loss == 0.0 and res.main_score == 0.0
    assert open(tmp_path / "x.tsv").read() == ""


def test_context_file_writer_matches_kb_context_process(tmp_path):
    """f-2: kbner.context_format.write_file against files produced by the reference's own kb/context_process.py
    `process_google` + `write_file` (tests/golden/context_format.json, oracle/gen_golden_context.py): the same retrieval
    dictionary and sub-token budget give byte-identical files -- contexts that do not fit are skipped while later shorter ones
    are taken, non-printable characters are dropped, a sentence without a hit is written bare, the scan stops below 10 free
    sub-tokens, and a train file (max_len = length_limit) drops sentences with more lines than the limit"""
    import tiny_assets
    from kbner import context_format as cf
    g = json.load(open(os.path.join(GOLD, "context_format.json"), encoding="utf-8"))
    tok = tiny_assets.build_tokenizer_dir(str(tmp_path / "tok"))
    count = lambda text: len(tok.tokenize(text))   # noqa: E731
    sents = []
    for s in g["sentences"]:
        toks = [tuple(line.split()) for line in s]
        key = " ".join(t[0] for t in toks).lower()
        sents.append({"tokens": toks, "contexts": g["google_dict"].get(key, [])})
    for run in g["runs"]:
        path = tmp_path / ("out_%d_%d.txt" % (run["length_limit"], run["max_len"]))
        cf.write_file(str(path), sents, count, length_limit=run["length_limit"], max_lines=run["max_len"])
        assert open(path, encoding="utf-8").read() == run["file"], (run["length_limit"], run["max_len"])
        st = cf.validate_file(str(path), count, length_limit=run["length_limit"], eos_text="</s>")
        assert st["over_budget"] == 0


def test_v2_doc_windows_match_reference(tmp_path):
    """f-3: TransformerWordEmbeddings(v2_doc=True): the document window cut around every sentence (ids, mask) and the positions
    of its first sub-tokens reproduce add_document_embeddings_v2's input and features (tests/golden/v2doc.npz, captured by
    running the reference): sentences at the document's ends get the other side's slack"""
    import tiny_assets
    from flair.data import Sentence
    from flair.embeddings import TransformerWordEmbeddings
    g = np.load(os.path.join(GOLD, "v2doc.npz"))
    mdir = tiny_assets.build_model_dir(str(tmp_path / "enc"), seed=0)
    tj = os.path.join(mdir, "tokenizer_config.json")
    cfg = json.load(open(tj))
    cfg["model_max_length"] = int(g["model_max_length"])
    json.dump(cfg, open(tj, "w"))
    emb = TransformerWordEmbeddings(model=mdir, layers="-1", pooling_operation="first", v2_doc=True)
    doc = [Sentence(str(t)) for t in g["texts"]]
    for i, s in enumerate(doc):
        s.doc, s.doc_pos = doc, i
    ids, am, first, lengths, first_row = emb.prepare_batch(doc)
    np.testing.assert_array_equal(ids, g["ids"])
    np.testing.assert_array_equal(am, g["mask"])
    assert [int(first[b, 0]) for b in range(len(doc))] == [int(p[0]) for p in g["batch_pos"]]
    hidden, feats = g["hidden"], g["features"]
    mine = np.zeros_like(feats)
    for b in range(len(doc)):
        for k in range(int(lengths[b])):
            if first[b, k] >= 0:
                mine[b, k] = hidden[first_row[b, k], first[b, k]]
    np.testing.assert_array_equal(mine, feats)


def test_sliding_windows_match_reference_run(tmp_path):
    """f-3: sentences longer than one encoder window.  tests/golden/windows.npz was captured by RUNNING the reference's
    TransformerWordEmbeddings (oracle/gen_golden_windows.py: the encode_plus overflow loop, flair/embeddings.py:3203-3227, and the
    seam stitching of window states, :3292-3299) with max_subtokens_sequence_length = 64, stride = 32 on one batch of sentences of
    1, 5, 9 and 19 windows.  The product's host side must build the same encoder rows and mask, and the (row, position) it gathers
    every word token from must be the element the reference's stitching reaches: checked here by indexing the REFERENCE's own
    hidden states with the product's index and comparing with the reference's features, bit for bit.
    The one deliberate deviation: transformers 3.0.0's `longest_first` loop returns the overflow scrambled (`ids_quirk`); the
    product implements the intended order -- the fixture holds both, and the quirk rows differ from the first overflow row on."""
    import tiny_assets
    from flair.data import Sentence
    from flair.embeddings import TransformerWordEmbeddings
    g = np.load(os.path.join(GOLD, "windows.npz"))
    mdir = tiny_assets.build_model_dir(str(tmp_path / "enc"), seed=0)
    emb = TransformerWordEmbeddings(model=mdir, layers="-1", pooling_operation="first")
    emb.max_subtokens_sequence_length, emb.stride, emb.allow_long_sentences = int(g["max_len"]), int(g["stride"]), True
    sents = [Sentence(str(t)) for t in g["texts"]]
    ids, am, first, lengths, first_row = emb.prepare_batch(sents)
    np.testing.assert_array_equal(ids, g["ids_intended"])
    np.testing.assert_array_equal(am, g["mask_intended"])
    assert list(lengths) == list(g["lengths"])
    rows = list(g["rows_per_sentence"])
    assert rows == [1, 5, 9, 19]
    r0 = np.concatenate([[0], np.cumsum(rows)])
    hidden, feats = g["hidden_intended"], g["features_intended"]
    mine = np.zeros_like(feats)
    for b in range(len(sents)):
        for k in range(int(lengths[b])):
            if first[b, k] >= 0:
                assert r0[b] <= first_row[b, k] < r0[b + 1]
                mine[b, k] = hidden[first_row[b, k], first[b, k]]
    np.testing.assert_array_equal(mine, feats)
    # the recorded deviation: same number of rows, identical up to (and including) the first window of every sentence, then
    # the literal 3.0.0 loop feeds the encoder scrambled ids
    q = g["ids_quirk"]
    assert q.shape == ids.shape
    for b in range(len(sents)):
        np.testing.assert_array_equal(q[r0[b]], ids[r0[b]])
    assert (q[r0[1] + 1] != ids[r0[1] + 1]).any()
