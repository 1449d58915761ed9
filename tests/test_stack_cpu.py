"""CPU: the oracle restatement of BASELINE config 5's inference stack (oracle/stack.py) and the mirror's host-side bookkeeping
against tests/golden/stack.* -- captured by running the reference's FastSequenceTagger(use_rnn=True) over two frozen
TransformerWordEmbeddings (one with use_internal_doc) + forward / backward FlairEmbeddings, three embedding-selection masks,
sentences chunked at <EOS> (oracle/gen_golden_stack.py)."""
import json
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


@pytest.fixture(scope="module")
def g13():
    return json.load(open(os.path.join(GOLD, "stack.json"))), np.load(os.path.join(GOLD, "stack.npz"))


def _rnn(z):
    return {k[len("w/rnn."):]: z[k] for k in z.files if k.startswith("w/rnn.")}


def test_lstm_restatement_equals_torch_lstm():
    """the oracle's LSTM equations == torch.nn.LSTM (the module the reference instantiates), packed bidirectional included"""
    from oracle import stack as ost
    torch.manual_seed(0)
    rnn = torch.nn.LSTM(12, 10, num_layers=1, bidirectional=True)
    x = torch.randn(3, 7, 12)
    lengths = [7, 3, 5]
    packed = torch.nn.utils.rnn.pack_padded_sequence(x.transpose(0, 1), lengths, enforce_sorted=False)
    out, _ = rnn(packed)
    ref, _ = torch.nn.utils.rnn.pad_packed_sequence(out, batch_first=True)
    sd = {k: v.detach().numpy() for k, v in rnn.state_dict().items()}
    mine = ost.bilstm_packed(x.numpy(), lengths, sd)
    np.testing.assert_allclose(mine, ref.detach().numpy(), rtol=1e-5, atol=1e-6)


def test_flair_features_match_the_reference(g13):
    from oracle import stack as ost
    meta, z = g13
    chars = ["<unk>"] + meta["lm_chars"]
    for bi, rec in enumerate(meta["batches"]):
        for tag, fwd in (("lm_f", True), ("lm_b", False)):
            lm = {k[len(tag) + 1:]: z[k] for k in z.files if k.startswith(tag + "/")}
            mine = ost.flair_features(rec["sentences"], lm, chars, fwd)
            np.testing.assert_allclose(mine, z["b%d/feat/%s" % (bi, tag)], rtol=2e-5, atol=2e-6)


def test_stack_emissions_match_the_reference(g13):
    """selection-masked concat in sorted-name order -> packed BiLSTM -> linear, for all three captured masks"""
    from oracle import stack as ost
    meta, z = g13
    rnn = _rnn(z)
    for bi, rec in enumerate(meta["batches"]):
        feats = {meta["names"][k]: z["b%d/feat/%s" % (bi, k)] for k in meta["names"]}
        lengths = [len(s) for s in rec["sentences"]]
        for si, sel in enumerate(rec["selections"]):
            em = ost.stack_emissions(feats, sel, lengths, rnn, z["w/linear.weight"], z["w/linear.bias"])
            ref = z["b%d/emissions/%d" % (bi, si)]
            for b, L in enumerate(lengths):
                np.testing.assert_allclose(em[b, :L], ref[b, :L], rtol=2e-4, atol=2e-5)


def test_viterbi_labels_of_the_stack_match_the_reference(g13):
    """oracle Viterbi on the reference's emissions == the reference's _obtain_labels output (tags and confidences)"""
    from oracle import crf as ocrf
    meta, z = g13
    items = meta["tag_dictionary"]
    start, stop = items.index("<START>"), items.index("<STOP>")
    for bi, rec in enumerate(meta["batches"]):
        lengths = np.asarray([len(s) for s in rec["sentences"]])
        for si in range(len(rec["selections"])):
            em = z["b%d/emissions/%d" % (bi, si)]
            tags, conf = ocrf.viterbi_batch(em, lengths, z["w/transitions"], start, stop)
            for b, L in enumerate(lengths):
                assert [items[t] for t in tags[b, :L]] == rec["labels"][si][b]
                np.testing.assert_allclose(conf[b, :L], rec["scores"][si][b], rtol=2e-5)


def test_mirror_char_batch_and_internal_doc_index(g13, tmp_path):
    """host bookkeeping of the mirror: FlairEmbeddings.char_batch picks the same (step, sequence) hidden states the reference
    does, and TransformerWordEmbeddings.prepare_stack_batch with use_internal_doc encodes the UNCHUNKED sentence while
    pooling only the chunked sentence's tokens"""
    import tiny_assets
    from oracle import stack as ost
    from flair.data import Dictionary, Sentence
    from flair.embeddings import FlairEmbeddings, TransformerWordEmbeddings
    from flair.models import LanguageModel
    meta, z = g13
    cd = Dictionary()
    for ch in meta["lm_chars"]:
        cd.add_item(ch)
    rec = meta["batches"][0]
    sents = []
    for toks, doc in zip(rec["sentences"], rec["doc_sentences"]):
        s = Sentence(" ".join(toks))
        s.doc_sent = Sentence(" ".join(doc))
        sents.append(s)
    n = max(len(s) for s in sents)
    for tag, fwd in (("lm_f", True), ("lm_b", False)):
        lm_sd = {k[len(tag) + 1:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(tag + "/")}
        lm = LanguageModel(cd, fwd, 48, 1, 20, None, 0.0, state_dict=lm_sd)
        lm.save(tmp_path / (tag + ".pt"))
        emb = FlairEmbeddings(str(tmp_path / (tag + ".pt")))
        assert emb.embedding_length == 48 and emb.is_forward_lm == fwd
        ids, rows = emb.char_batch(sents, n)
        # run the oracle LSTM over the mirror's ids and read the states at the mirror's (step, sequence) -> row table
        lmn = {k: v.numpy() for k, v in lm_sd.items()}
        hs, _ = ost.lstm_layer(lmn["encoder.weight"][ids], lmn["rnn.weight_ih_l0"], lmn["rnn.weight_hh_l0"], lmn["rnn.bias_ih_l0"],
                               lmn["rnn.bias_hh_l0"])
        mine = np.zeros((len(sents), n, 48), np.float32)
        st, bb = np.nonzero(rows >= 0)
        for s_, b_ in zip(st, bb):
            r = rows[s_, b_]
            mine[r // n, r % n] = hs[s_, b_]
        np.testing.assert_allclose(mine, z["b0/feat/" + tag], rtol=2e-5, atol=2e-6)
    mdir = tiny_assets.build_model_dir(str(tmp_path / "enc_a"), seed=0)
    emb = TransformerWordEmbeddings(model=mdir, layers="-1", pooling_operation="first", use_internal_doc=True)
    ids, am, first, lengths, first_row = emb.prepare_stack_batch(sents)
    ids_doc = emb.prepare_batch([s.doc_sent for s in sents])[0]
    np.testing.assert_array_equal(ids, ids_doc)                       # the encoder reads the unchunked sentence
    assert first.shape == (len(sents), n) and lengths.tolist() == [len(s) for s in sents]
    assert all((first[b, :len(s)] >= 0).all() and (first[b, len(s):] < 0).all() for b, s in enumerate(sents))


def test_bert_embeddings_index_matches_reference_features(tmp_path):
    """G14: the mirror's BertEmbeddings.prepare_stack_batch feeds the encoder the reference's ids / mask, and gathering the
    reference's hidden states (layers -1..-4) at the mirror's first-piece positions reproduces the reference's features --
    including the token made of a control character, which gets no piece and borrows the next token's first piece"""
    import tiny_assets
    from flair.data import Sentence
    from flair.embeddings import BertEmbeddings
    g = np.load(os.path.join(GOLD, "bert_embeddings.npz"))
    mdir = tiny_assets.build_bert_dir(str(tmp_path / "bert-tiny"), seed=9)
    emb = BertEmbeddings(bert_model_or_path=mdir, layers="-1,-2,-3,-4", pooling_operation="first")
    assert emb.embedding_length == int(g["embedding_length"])
    sents = [Sentence(str(t)) for t in g["texts"]]
    assert [len(s) for s in sents] == g["n_tokens"].tolist()
    ids, am, first, lengths = emb.prepare_stack_batch(sents)
    np.testing.assert_array_equal(ids, g["ids"])
    np.testing.assert_array_equal(am, g["mask"])
    hs = [g["hs%d" % i] for i in range(5)]
    feats = g["features"]
    mine = np.zeros_like(feats)
    for b in range(len(sents)):
        for k in range(int(lengths[b])):
            mine[b, k] = np.concatenate([hs[len(hs) + li][b, first[b, k]] for li in emb.layer_indexes])
    np.testing.assert_array_equal(mine, feats)
