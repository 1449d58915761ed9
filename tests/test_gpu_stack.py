"""GPU: BASELINE config 5's inference stack on the HIP engine (kbner.stack + csrc/lstm.hip) -- frozen stacked embeddings
(two XLM-R-shaped encoders, one with use_internal_doc; forward + backward character LMs) -> selection-masked concat -> BiLSTM ->
linear -> CRF Viterbi -- against tests/golden/stack.* captured by running the reference (oracle/gen_golden_stack.py), and the
LSTM step kernel alone against the oracle restatement."""
import json
import os

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


@pytest.fixture(scope="module")
def g13():
    return json.load(open(os.path.join(GOLD, "stack.json"))), np.load(os.path.join(GOLD, "stack.npz"))


@pytest.fixture(scope="module")
def assets(tmp_path_factory, g13):
    import tiny_assets
    from flair.data import Dictionary
    from flair.models import LanguageModel
    meta, z = g13
    d = tmp_path_factory.mktemp("stack")
    tiny_assets.build_model_dir(str(d / "enc_a"), seed=0)
    tiny_assets.build_model_dir(str(d / "enc_b"), seed=5)
    cd = Dictionary()
    for ch in meta["lm_chars"]:
        cd.add_item(ch)
    for tag, fwd in (("lm_f", True), ("lm_b", False)):
        sd = {k[len(tag) + 1:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(tag + "/")}
        LanguageModel(cd, fwd, 48, 1, 20, None, 0.0, state_dict=sd).save(d / (tag + ".pt"))
    return d


def test_char_lm_group_matches_single_models():
    """kbner_lstm_seq with several character LMs as ONE group (stacked tables, per-model ids and output columns that are NOT
    equidistant) == each model run alone through the one-wave step kernel (kbner_lstm_step), and the final states agree"""
    from kbner import stack as K
    rng = np.random.default_rng(7)
    g = torch.Generator().manual_seed(5)
    H, B, steps, n = 96, 21, 23, 4
    lms = []
    for i in range(3):
        k = 1.0 / H ** 0.5
        chars = 40 + 7 * i          # dictionaries of different sizes
        sd = {"encoder.weight": torch.rand(chars, 20, generator=g) * 0.4 - 0.2,
              "rnn.weight_ih_l0": (torch.rand(4 * H, 20, generator=g) * 2 - 1) * k,
              "rnn.weight_hh_l0": (torch.rand(4 * H, H, generator=g) * 2 - 1) * k * 3,
              "rnn.bias_ih_l0": torch.rand(4 * H, generator=g) * 0.2, "rnn.bias_hh_l0": torch.rand(4 * H, generator=g) * 0.2}
        lms.append(K.CharLM(sd, H, "cuda"))
    ids = [rng.integers(0, 40 + 7 * i, size=(steps, B)).astype(np.int32) for i in range(3)]
    rows = []
    for i in range(3):
        r = np.full((steps, B), -1, np.int32)
        for b in range(B):
            for t, s in enumerate(sorted(rng.choice(np.arange(1, steps), size=n, replace=False))):
                r[s, b] = b * n + t
        rows.append(r)
    cols = [0, 128, 352]            # not equidistant
    ld = 512
    X1 = torch.zeros((128, ld), dtype=torch.bfloat16, device="cuda")
    X2 = torch.zeros((128, ld), dtype=torch.bfloat16, device="cuda")
    K.CharLMGroup(lms).run(ids, rows, X1, cols)
    for i, lm in enumerate(lms):
        gxi = torch.from_numpy(ids[i][:, None, :].copy()).cuda()
        outi = torch.from_numpy(rows[i][:, None, :].copy()).cuda()
        lm.grp.run_stepwise(lm.table, gxi, outi, X2, 0, B, col=cols[i])
    torch.cuda.synchronize()
    a, b = X1.float().cpu().numpy(), X2.float().cpu().numpy()
    assert np.abs(b).max() > 0.05
    # same bf16 inputs, fp32 accumulation in a different order (4 K-slices summed through LDS): a few bf16 ulps after 23 steps
    assert np.abs(a - b).max() <= 2e-2 * np.abs(b).max(), np.abs(a - b).max()
    assert (np.abs(a) > 0).sum() == (np.abs(b) > 0).sum() == 3 * B * n * H


@pytest.mark.parametrize("B,n,D,H", [(3, 7, 40, 24), (5, 19, 96, 100), (33, 12, 64, 32), (4, 9, 128, 1000),   # config 5's hidden_size 1000 -> 1024
                                     (20, 6, 64, 96), (50, 5, 64, 64), (70, 4, 64, 32)])   # 2 / 4 sequence tiles per workgroup, two batch chunks
def test_bilstm_head_vs_oracle(B, n, D, H):
    """input GEMM + per-step recurrence kernel (both directions, ragged lengths, hidden padded to 32) + linear vs the numpy LSTM"""
    from kbner import stack as K
    from oracle import stack as ost
    rng = np.random.default_rng(B * 100 + n)
    T = 9
    ws = 0.3 if H <= 100 else 1.5 / H ** 0.5     # keep the gates out of saturation at the real hidden size (torch's init is U(+-1/sqrt(H)))
    rnn = {}
    for sfx in ("", "_reverse"):
        rnn["weight_ih_l0" + sfx] = rng.standard_normal((4 * H, D)).astype(np.float32) * ws
        rnn["weight_hh_l0" + sfx] = rng.standard_normal((4 * H, H)).astype(np.float32) * ws
        rnn["bias_ih_l0" + sfx] = rng.standard_normal(4 * H).astype(np.float32) * 0.2
        rnn["bias_hh_l0" + sfx] = rng.standard_normal(4 * H).astype(np.float32) * 0.2
    lw = rng.standard_normal((T, 2 * H)).astype(np.float32) * ws
    lb = rng.standard_normal(T).astype(np.float32)
    lengths = rng.integers(1, n + 1, size=B)
    lengths[0] = n
    blocks = [D - 8, 8] if D > 8 else [D]     # two feature blocks: the second starts at a 32-aligned column of X
    head = K.BiLSTMHead({k: torch.from_numpy(v) for k, v in rnn.items()}, lw, lb, blocks, H, "cuda")
    x = rng.standard_normal((B, n, D)).astype(np.float32)
    for b in range(B):
        x[b, lengths[b]:] = 0
    xb = torch.from_numpy(x).to(torch.bfloat16)
    X = head.new_input(B, n)
    ref_off = 0
    for w, c in zip(blocks, head.cols):
        X[:B * n, c:c + w] = xb.view(B * n, D)[:, ref_off:ref_off + w].cuda()
        ref_off += w
    em = head.emissions(X, lengths, B, n).cpu().numpy()
    ref = ost.stack_emissions({"a": xb.float().numpy()}, [1], lengths, rnn, lw, lb)
    num = den = 0.0
    for b in range(B):
        num += float(((em[b, :lengths[b]] - ref[b, :lengths[b]]) ** 2).sum())
        den += float((ref[b, :lengths[b]] ** 2).sum())
    rel = (num / den) ** 0.5
    print("bilstm head rel L2", (B, n, D, H), rel)
    assert rel < 1.1e-2, rel          # bf16 weights / hidden states vs fp32; observed <= 3.5e-3


def test_stack_tagger_vs_reference_golden(assets, g13):
    """the mirror's FastSequenceTagger(use_rnn=True) with the reference run's weights: emissions within bf16 tolerance of the
    reference's fp32 emissions for every selection mask, Viterbi labels identical (a near-tie may flip: at most 1 in 50)"""
    from flair.custom_data_loader import BatchedData
    from flair.data import Dictionary, Sentence
    from flair.embeddings import FlairEmbeddings, StackedEmbeddings, TransformerWordEmbeddings
    from flair.models import FastSequenceTagger
    meta, z = g13
    d = assets
    td = Dictionary(add_unk=False)
    for it in meta["tag_dictionary"]:
        td.add_item(it)
    embs = [TransformerWordEmbeddings(model=str(d / "enc_a"), layers="-1", pooling_operation="first", use_internal_doc=True),
            TransformerWordEmbeddings(model=str(d / "enc_b"), layers="-1", pooling_operation="first"),
            FlairEmbeddings(str(d / "lm_f.pt")), FlairEmbeddings(str(d / "lm_b.pt"))]
    tagger = FastSequenceTagger(hidden_size=40, embeddings=StackedEmbeddings(embs), tag_dictionary=td, tag_type="ner", use_crf=True,
                                use_rnn=True, dropout=0.0, word_dropout=0.05, locked_dropout=0.5, sentence_loss=True, remove_x=True,
                                embedding_selector=True, use_rl=True)
    assert [os.path.basename(e.name) for e in tagger._stack_embs] == [os.path.basename(nm) for nm in meta["sorted_names"]]
    tagger.load_stack_state({"rnn": {k[len("w/rnn."):]: z[k] for k in z.files if k.startswith("w/rnn.")},
                             "linear.weight": z["w/linear.weight"], "linear.bias": z["w/linear.bias"],
                             "transitions": z["w/transitions"]})
    tagger.eval()
    tot = flips = 0
    worst = 0.0
    for bi, rec in enumerate(meta["batches"]):
        sents = []
        for toks, doc, tg in zip(rec["sentences"], rec["doc_sentences"], rec["tags"]):
            s = Sentence(" ".join(toks))
            s.doc_sent = Sentence(" ".join(doc))
            for t, v in zip(s, tg):
                t.add_tag("ner", v)
            sents.append(s)
        batch = BatchedData(sents)
        lengths = [len(s) for s in sents]
        for si, sel in enumerate(rec["selections"]):
            tagger.selection = torch.tensor(sel)
            feats = tagger.forward(batch)
            em, ref = feats.cpu().numpy(), z["b%d/emissions/%d" % (bi, si)]
            num = sum(float(((em[b, :L] - ref[b, :L]) ** 2).sum()) for b, L in enumerate(lengths))
            den = sum(float((ref[b, :L] ** 2).sum()) for b, L in enumerate(lengths))
            worst = max(worst, (num / den) ** 0.5)
            labels, _ = tagger._obtain_labels(feats, batch)
            for b, row in enumerate(labels):
                assert len(row) == lengths[b]
                for lab, want, sc in zip(row, rec["labels"][si][b], rec["scores"][si][b]):
                    tot += 1
                    if lab.value != want:
                        flips += 1
                    else:
                        assert abs(lab.score - sc) < 5e-2
    print("stack emissions worst rel L2", worst, "labels", tot, "flips", flips)
    assert worst < 2.1e-2, worst      # observed 6.9e-3 (114 labels, 0 flips)
    assert flips <= max(1, tot // 50), (flips, tot)


def test_bert_embeddings_device_features_vs_reference_golden(tmp_path):
    """G14 on the device: a BERT-shaped encoder (absolute positions, token-type row 0, LayerNorm eps 1e-12) on the HIP engine,
    last four layers of every token's first piece written into X == the reference BertEmbeddings' features (bf16 tolerance)"""
    import tiny_assets
    from flair.data import Dictionary, Sentence
    from flair.embeddings import BertEmbeddings, StackedEmbeddings
    from flair.models import FastSequenceTagger
    g = np.load(os.path.join(GOLD, "bert_embeddings.npz"))
    mdir = tiny_assets.build_bert_dir(str(tmp_path / "bert-tiny"), seed=9)
    emb = BertEmbeddings(bert_model_or_path=mdir, layers="-1,-2,-3,-4", pooling_operation="first")
    td = Dictionary(add_unk=False)
    for it in ("<unk>", "O", "S-PER", "<START>", "<STOP>"):
        td.add_item(it)
    tagger = FastSequenceTagger(hidden_size=32, embeddings=StackedEmbeddings([emb]), tag_dictionary=td, tag_type="ner", use_crf=True,
                                use_rnn=True, dropout=0.0)
    sents = [Sentence(str(t)) for t in g["texts"]]
    X, lengths, B, n = tagger._stack_input(sents)
    D = emb.embedding_length
    mine = X[:B * n, :D].float().cpu().numpy().reshape(B, n, D)
    ref = g["features"]
    num = sum(float(((mine[b, :L] - ref[b, :L]) ** 2).sum()) for b, L in enumerate(lengths))
    den = sum(float((ref[b, :L] ** 2).sum()) for b, L in enumerate(lengths))
    rel = (num / den) ** 0.5
    print("bert features rel L2", rel)
    assert rel < 2e-2, rel
    assert not mine[1, lengths[1]:].any() and not mine[2, lengths[2]:].any()      # padding rows stay zero


def test_sliding_window_features_vs_reference_run(tmp_path):
    """f-3 on the device: a frozen TransformerWordEmbeddings with 64-position windows (stride 32) over sentences of 1, 5, 9 and 19
    windows -- the encoder runs on every window row and gather_rows_ld pulls each word token's first sub-token from its
    (row, position) == the features the REFERENCE assigned after stitching its window states (tests/golden/windows.npz,
    oracle/gen_golden_windows.py), within the bf16 tolerance of the encoder"""
    import tiny_assets
    from flair.data import Dictionary, Sentence
    from flair.embeddings import StackedEmbeddings, TransformerWordEmbeddings
    from flair.models import FastSequenceTagger
    g = np.load(os.path.join(GOLD, "windows.npz"))
    mdir = tiny_assets.build_model_dir(str(tmp_path / "enc"), seed=0)
    emb = TransformerWordEmbeddings(model=mdir, layers="-1", pooling_operation="first", fine_tune=False)
    emb.max_subtokens_sequence_length, emb.stride, emb.allow_long_sentences = int(g["max_len"]), int(g["stride"]), True
    td = Dictionary(add_unk=False)
    for it in ("<unk>", "O", "S-PER", "<START>", "<STOP>"):
        td.add_item(it)
    tagger = FastSequenceTagger(hidden_size=32, embeddings=StackedEmbeddings([emb]), tag_dictionary=td, tag_type="ner", use_crf=True,
                                use_rnn=True, dropout=0.0)
    sents = [Sentence(str(t)) for t in g["texts"]]
    X, lengths, B, n = tagger._stack_input(sents)
    D = emb.embedding_length
    mine = X[:B * n, :D].float().cpu().numpy().reshape(B, n, D)
    ref = g["features_intended"]
    assert list(lengths) == list(g["lengths"])
    num = sum(float(((mine[b, :L] - ref[b, :L]) ** 2).sum()) for b, L in enumerate(lengths))
    den = sum(float((ref[b, :L] ** 2).sum()) for b, L in enumerate(lengths))
    rel = (num / den) ** 0.5
    print("sliding-window features rel L2", rel)
    assert rel < 2e-2, rel
    # per sentence too: a wrong seam would spoil one sentence's tail, not the batch average
    for b, L in enumerate(lengths):
        rb = (float(((mine[b, :L] - ref[b, :L]) ** 2).sum()) / float((ref[b, :L] ** 2).sum())) ** 0.5
        assert rb < 3e-2, (b, rb)


def test_config5_yaml_route(assets, tmp_path):
    """the ACE-shaped YAML through ConfigParser -> ReinforcementTrainer (assign_doc_for_ext_context chunks every sentence at
    <EOS>) -> train.py's --parse steps: selection mask set on the student, evaluate() writes one line per REAL token"""
    import tiny_assets
    import flair
    from flair.config_parser import ConfigParser
    from flair.custom_data_loader import ColumnDataLoader
    from flair.utils.from_params import Params
    d = assets
    tiny_assets.write_conll_corpus(str(tmp_path / "data"), n_train=6, n_dev=3, n_test=5, seed=3)
    cfg = {"ReinforcementTrainer": {"assign_doc_for_ext_context": True, "controller_learning_rate": 0.1, "controller_optimizer": "SGD",
                                    "distill_mode": False, "optimizer": "SGD", "sentence_level_batch": True},
           "embeddings": {"ELMoEmbeddings-0": {"options_file": "elmo/elmo_2x4096_512_2048cnn_2xhighway_5.5B_options.json",
                                               "weight_file": "elmo/elmo_2x4096_512_2048cnn_2xhighway_5.5B_weights.hdf5"},
                          "FastWordEmbeddings-0": {"embeddings": "en", "freeze": True},
                          "FlairEmbeddings-0": {"model": str(d / "lm_f.pt")}, "FlairEmbeddings-1": {"model": str(d / "lm_b.pt")},
                          "TransformerWordEmbeddings-0": {"layers": "-1", "model": str(d / "enc_a"), "pooling_operation": "first",
                                                          "use_internal_doc": True},
                          "TransformerWordEmbeddings-1": {"layers": "-1", "model": str(d / "enc_b"), "pooling_operation": "first"}},
           "model": {"FastSequenceTagger": {"crf_attention": False, "dropout": 0.0, "hidden_size": 40, "remove_x": True,
                                            "sentence_loss": True, "use_crf": True}},
           "model_name": "ace_tiny", "target_dir": str(tmp_path / "out"), "targets": "ner", "trainer": "ReinforcementTrainer",
           "ner": {"Corpus": "ColumnCorpus-TINY", "tag_dictionary": str(tmp_path / "tags.pkl"),
                   "ColumnCorpus-TINY": {"column_format": {0: "text", 1: "pos", 2: "upos", 3: "ner"}, "comment_symbol": "# id",
                                         "data_folder": str(tmp_path / "data"), "tag_to_bioes": "ner"}},
           "train": {"controller_momentum": 0.9, "learning_rate": 0.1, "max_episodes": 30, "max_epochs": 150, "mini_batch_size": 32,
                     "monitor_test": False, "train_with_dev": False}}
    with open(tmp_path / "cfg.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    cp = ConfigParser(Params.from_file(str(tmp_path / "cfg.yaml")))
    student = cp.create_student()
    assert student.use_rnn and student.stack_head is not None
    trainer = getattr(flair.trainers, cp.config["trainer"])(student, None, cp.corpus, config=cp.config,
                                                             **cp.config["ReinforcementTrainer"], is_test=True)
    assert student.embedding_selector and student.use_rl
    for s in cp.corpus.test_list[0]:
        assert "<EOS>" not in [t.text for t in s] and "<EOS>" in [t.text for t in s.doc_sent]
    # the shipped ACE YAML also lists ELMo and fastText embeddings, whose weights this offline build cannot have: they are
    # placeholders (reference name + width, so the BiLSTM's input layout is the trained one) that work only DESELECTED
    names = [e.name for e in student._stack_embs]
    assert names == sorted(names) and "elmo-original" in names and "Word: en" in names
    assert student.embeddings.embedding_length == 3072 + 300 + sum(e.embedding_length for e in student._stack_embs
                                                                   if e.name not in ("elmo-original", "Word: en"))
    sel = [0 if n in ("elmo-original", "Word: en") else 1 for n in names]
    sel[[i for i, n in enumerate(names) if n not in ("elmo-original", "Word: en")][1]] = 0   # + one real embedding off
    student.selection = sel                    # what train.py:216-217 loads from training_state.pt['best_action']
    student.eval()
    loader = ColumnDataLoader(list(trainer.corpus.test), 4, use_bert=student.use_bert, model=student, sort_data=False,
                              sentence_level_batch=True)
    loader.assign_tags(student.tag_type, student.tag_dictionary)
    res, loss = student.evaluate(loader, out_path=tmp_path / "pred.conllu", embeddings_storage_mode="none", prediction_mode=True)
    lines = [l for l in open(tmp_path / "pred.conllu").read().split("\n") if l]
    assert len(lines) == sum(len(s) for s in cp.corpus.test_list[0])
    assert all(len(l.split(" ")) == 4 and l.split(" ")[1] != "S-X" for l in lines)
    assert np.isfinite(loss) and 0.0 <= res.main_score <= 1.0
    # the same through the command line (train.py --test): best_action from training_state.pt, head from stack-model.pt
    import subprocess
    import sys
    base = cp.get_target_path
    os.makedirs(base, exist_ok=True)
    torch.save({"episode": 3, "best_action": sel}, str(base / "training_state.pt"))
    torch.save(student._stack_state, str(base / "stack-model.pt"))
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "train.py"), "--config", str(tmp_path / "cfg.yaml"), "--test", "--keep_order"],
                       cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    cli = [l.split(" ") for l in open(base / "ColumnCorpus-TINY-test.tsv").read().split("\n") if l]
    mine = [l.split(" ") for l in lines]
    assert len(cli) == len(mine) and [c[:3] for c in cli] == [m[:3] for m in mine]    # same tokens, gold, predicted tags
    r = subprocess.run([sys.executable, os.path.join(root, "train.py"), "--config", str(tmp_path / "cfg.yaml")],
                       cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode != 0 and b"out of scope" in r.stderr + r.stdout               # training an ACE YAML is refused, loudly
    student.selection = [1] * len(names)
    with pytest.raises(NotImplementedError, match="not available offline|no device producer"):
        student.evaluate(loader, embeddings_storage_mode="none", prediction_mode=True)   # a selected placeholder refuses
    with pytest.raises(NotImplementedError):
        trainer.train(cp.get_target_path, **cp.config["train"])      # ACE controller training is out of scope
