"""Test helpers: a tiny local 'HF directory' (tokenizer + config + random-init weights) and a KB-NER-style CoNLL corpus,
built at test time -- there are no XLM-R weights / vocab on this machine and no network."""
import json
import os

import numpy as np

WORDS = ("the quick brown fox jumps over lazy dog berlin paris london zalando research amazon google university of "
         "cambridge river thames eiffel tower alice bob carol visited works at lives in near and or with from to new york "
         "city museum art science music band album song released year 1999 2021 company founded by john smith mary jones "
         "wikipedia article about history location country capital population language people famous known for").split()


def build_tokenizer_dir(path, vocab_size=300, seed=0):
    """Unigram + Metaspace fast tokenizer with <s>=0 <pad>=1 </s>=2 <unk>=3 (XLM-R's special-token ids), saved so that
    AutoTokenizer.from_pretrained(path) loads it."""
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, trainers, decoders
    from transformers import PreTrainedTokenizerFast
    rng = np.random.default_rng(seed)
    corpus = [" ".join(rng.choice(WORDS, size=12)) for _ in range(400)]
    tok = Tokenizer(models.Unigram())
    # like XLM-R's sentencepiece normaliser (nmt_nfkc), which deletes soft hyphens / zero-width characters: a word token made
    # only of such characters receives NO sub-token (the 0-length case of flair/embeddings.py:3306-3308)
    tok.normalizer = normalizers.Replace("\u00ad", "")
    tok.pre_tokenizer = pre_tokenizers.Metaspace()
    tok.decoder = decoders.Metaspace()
    tr = trainers.UnigramTrainer(vocab_size=vocab_size, special_tokens=["<s>", "<pad>", "</s>", "<unk>"], unk_token="<unk>")
    tok.train_from_iterator(corpus, tr)
    os.makedirs(path, exist_ok=True)
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, bos_token="<s>", eos_token="</s>", pad_token="<pad>", unk_token="<unk>",
                                   cls_token="<s>", sep_token="</s>", model_max_length=512)
    fast.save_pretrained(path)
    return fast


def build_model_dir(path, hidden=128, layers=2, heads=2, inter=256, seed=0, tokenizer="unigram"):
    """tokenizer + config.json + model.safetensors (random init, HF names) for a tiny XLM-R-shaped encoder (head_dim 64)"""
    import torch
    from safetensors.torch import save_file
    tok = build_tokenizer_dir(path, seed=seed) if tokenizer == "unigram" else build_wordpiece_tokenizer_dir(path, seed=seed)
    V = len(tok)
    cfg = dict(model_type="xlm-roberta", architectures=["XLMRobertaModel"], vocab_size=V, hidden_size=hidden,
               num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=inter, max_position_embeddings=514,
               type_vocab_size=1, pad_token_id=1, bos_token_id=0, eos_token_id=2, layer_norm_eps=1e-5, hidden_act="gelu",
               hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f)
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def w(*shape, std=0.05):
        return torch.empty(*shape).normal_(0, std, generator=g)

    sd["embeddings.word_embeddings.weight"] = w(V, hidden)
    sd["embeddings.position_embeddings.weight"] = w(514, hidden)
    sd["embeddings.token_type_embeddings.weight"] = w(1, hidden)
    sd["embeddings.LayerNorm.weight"] = torch.ones(hidden)
    sd["embeddings.LayerNorm.bias"] = torch.zeros(hidden)
    for i in range(layers):
        p = "encoder.layer.%d." % i
        for nm in ("query", "key", "value"):
            sd[p + "attention.self.%s.weight" % nm] = w(hidden, hidden)
            sd[p + "attention.self.%s.bias" % nm] = torch.zeros(hidden)
        sd[p + "attention.output.dense.weight"] = w(hidden, hidden)
        sd[p + "attention.output.dense.bias"] = torch.zeros(hidden)
        sd[p + "attention.output.LayerNorm.weight"] = torch.ones(hidden)
        sd[p + "attention.output.LayerNorm.bias"] = torch.zeros(hidden)
        sd[p + "intermediate.dense.weight"] = w(inter, hidden)
        sd[p + "intermediate.dense.bias"] = torch.zeros(inter)
        sd[p + "output.dense.weight"] = w(hidden, inter)
        sd[p + "output.dense.bias"] = torch.zeros(hidden)
        sd[p + "output.LayerNorm.weight"] = torch.ones(hidden)
        sd[p + "output.LayerNorm.bias"] = torch.zeros(hidden)
    save_file(sd, os.path.join(path, "model.safetensors"), metadata={"format": "pt"})
    return path


ENTITIES = {"berlin": "LOC", "paris": "LOC", "london": "LOC", "zalando": "CORP", "amazon": "CORP", "google": "CORP",
            "alice": "PER", "bob": "PER", "carol": "PER"}


def write_conll_corpus(folder, n_train=24, n_dev=8, n_test=8, seed=0):
    """KB-NER file format (kb/context_process.py output): `token POS UPOS NER` columns, `# id ...` comment lines, the sentence,
    then `<EOS> B-X B-X B-X` and retrieved-context tokens all tagged `B-X`."""
    rng = np.random.default_rng(seed)
    os.makedirs(folder, exist_ok=True)

    def sentence():
        n = int(rng.integers(4, 9))
        words = list(rng.choice(WORDS, size=n))
        if rng.random() < 0.9:
            words[int(rng.integers(0, n))] = str(rng.choice(list(ENTITIES)))
        lines = []
        for wd in words:
            tag = "B-" + ENTITIES[wd] if wd in ENTITIES else "O"
            lines.append("%s _ _ %s" % (wd, tag))
        lines.append("<EOS> B-X B-X B-X")
        for wd in rng.choice(WORDS, size=int(rng.integers(5, 14))):
            lines.append("%s B-X B-X B-X" % wd)
        return lines

    for name, k in (("train.txt", n_train), ("dev.txt", n_dev), ("test.txt", n_test)):
        with open(os.path.join(folder, name), "w") as f:
            for i in range(k):
                f.write("# id %s-%d\tdomain=en\n" % (name, i))
                f.write("\n".join(sentence()) + "\n\n")
    return folder


def write_multiview_corpora(folder_plain, folder_doc, n_train=12, n_dev=4, n_test=4, seed=5):
    """the two views of a multi-view (cooperative-learning) run, sentence k of one file being sentence k of the other: the bare
    sentences (`token _ _ NER`) and the same sentences followed by `<EOS>` + retrieved context tagged B-X (the *DOC corpus)"""
    rng = np.random.default_rng(seed)
    os.makedirs(folder_plain, exist_ok=True)
    os.makedirs(folder_doc, exist_ok=True)
    for name, k in (("train.txt", n_train), ("dev.txt", n_dev), ("test.txt", n_test)):
        with open(os.path.join(folder_plain, name), "w") as fp, open(os.path.join(folder_doc, name), "w") as fd:
            for i in range(k):
                n = int(rng.integers(3, 8))
                words = list(rng.choice(WORDS, size=n))
                if rng.random() < 0.9:
                    words[int(rng.integers(0, n))] = str(rng.choice(list(ENTITIES)))
                sent = ["%s _ _ %s" % (wd, "B-" + ENTITIES[wd] if wd in ENTITIES else "O") for wd in words]
                ctx = ["<EOS> B-X B-X B-X"] + ["%s B-X B-X B-X" % wd for wd in rng.choice(WORDS, size=int(rng.integers(4, 10)))]
                for f, lines in ((fp, sent), (fd, sent + ctx)):
                    f.write("# id %s-%d\tdomain=en\n" % (name, i))
                    f.write("\n".join(lines) + "\n\n")


def multiview_config(d, max_epochs=3, accum=2, mini_batch_size=2, temperature=4.0, n_train=12, n_dev=4, n_test=4):
    """e2e_config with the shape of the shipped *_doc_joint_multiview_posterior_* YAMLs: a plain corpus and its *DOC twin
    trained jointly, multi_view_training + distill_posterior + temperature on the tagger (no dropout, no shuffling, so the
    reference's run and the mirror's can be compared step by step)"""
    cfg = e2e_config(d, word_dropout=0.0, max_epochs=max_epochs, shuffle=False, accum=accum, mini_batch_size=mini_batch_size,
                     save_finetuned_embedding=False)
    d = str(d)
    write_multiview_corpora(os.path.join(d, "mv_plain"), os.path.join(d, "mv_doc"), n_train=n_train, n_dev=n_dev, n_test=n_test)
    fmt = {"column_format": {0: "text", 1: "pos", 2: "upos", 3: "ner"}, "comment_symbol": "# id", "tag_to_bioes": "ner"}
    cfg["ner"] = {"Corpus": "ColumnCorpus-TINYMV:ColumnCorpus-TINYMVDOC", "tag_dictionary": os.path.join(d, "tags_mv.pkl"),
                  "ColumnCorpus-TINYMV": dict(fmt, data_folder=os.path.join(d, "mv_plain")),
                  "ColumnCorpus-TINYMVDOC": dict(fmt, data_folder=os.path.join(d, "mv_doc"))}
    cfg["model"]["FastSequenceTagger"].update(multi_view_training=True, distill_posterior=True, temperature=temperature)
    cfg["model_name"] = "tiny_mv_run"
    return cfg


def kd_config(d, max_epochs=2, accum=2, mini_batch_size=3, temperature=2.0, interpolation=0.5, best_k=3, n_train=18, n_dev=6,
              n_test=4, posterior=True, crf=True, attention=True, exact=False):
    """teacher-student knowledge distillation on the tiny corpus, the way the reference configures it: `ModelFinetuner:
    {distill_mode: true}`, the KD switches on the student model, `interpolation` at the top level, `is_teacher_list: true` +
    `ner.teachers: {<teacher yaml>: <corpora it teaches>}` (flair/config_parser.py:255-274).  -> (student config, teacher config);
    the teacher's YAML is the plain e2e config under its own model_name (its best-model.pt is written by the caller)."""
    d = str(d)
    kw = dict(word_dropout=0.0, max_epochs=max_epochs, shuffle=False, accum=accum, mini_batch_size=mini_batch_size, n_train=n_train,
              n_dev=n_dev, n_test=n_test, save_finetuned_embedding=False)
    teacher = e2e_config(d, **kw)
    teacher["model_name"] = "tiny_teacher"
    cfg = e2e_config(d, **kw)
    cfg["model_name"] = "tiny_kd_run"
    cfg["ModelFinetuner"]["distill_mode"] = True
    cfg["model"]["FastSequenceTagger"].update(distill_posterior=posterior, distill_crf=crf, crf_attention=attention,
                                              distill_exact=exact, temperature=temperature)
    cfg["interpolation"] = interpolation
    cfg["is_teacher_list"] = True
    cfg["ner"]["teachers"] = {os.path.join(d, "teacher.yaml"): "ColumnCorpus-TINY"}
    cfg["train"]["best_k"] = best_k
    return cfg, teacher


def e2e_config(d, word_dropout=0.1, max_epochs=6, shuffle=None, n_train=32, n_dev=8, n_test=8, accum=2, mini_batch_size=4,
               save_finetuned_embedding=True):
    """The KB-NER-shaped YAML (as a dict) of the tiny end-to-end run under directory `d`: builds the model dir + corpus files
    and returns the config.  Shared by tests/test_gpu_flair_e2e.py and oracle/gen_golden_e2e.py (G12), so the reference and the
    mirror are configured by the same keys."""
    d = str(d)
    build_model_dir(os.path.join(d, "xlmr-tiny"))
    write_conll_corpus(os.path.join(d, "data"), n_train=n_train, n_dev=n_dev, n_test=n_test)
    cfg = {
        "ModelFinetuner": {"distill_mode": False, "sentence_level_batch": True},
        "embeddings": {"TransformerWordEmbeddings-0": {"fine_tune": True, "layers": "-1", "model": os.path.join(d, "xlmr-tiny"),
                                                        "pooling_operation": "first"}},
        "model": {"FastSequenceTagger": {"crf_attention": False, "dropout": 0.0, "hidden_size": 256, "locked_dropout": 0.0,
                                         "remove_x": True, "sentence_loss": True, "use_cnn": False, "use_crf": True,
                                         "use_rnn": False, "word_dropout": word_dropout}},
        "model_name": "tiny_run", "target_dir": os.path.join(d, "out"), "targets": "ner", "trainer": "ModelFinetuner",
        "ner": {"Corpus": "ColumnCorpus-TINY", "tag_dictionary": os.path.join(d, "tags.pkl"),
                "ColumnCorpus-TINY": {"column_format": {0: "text", 1: "pos", 2: "upos", 3: "ner"}, "comment_symbol": "# id",
                                      "data_folder": os.path.join(d, "data"), "tag_to_bioes": "ner"}},
        "train": {"embeddings_storage_mode": "none", "fine_tune_mode": True, "gradient_accumulation_steps": accum,
                  "learning_rate": 2.0e-3, "lr_rate": 50, "max_epochs": max_epochs, "mini_batch_size": mini_batch_size,
                  "monitor_test": False, "save_finetuned_embedding": save_finetuned_embedding, "select_model_by_macro": True,
                  "train_with_dev": False, "true_reshuffle": False, "use_warmup": False},
    }
    if shuffle is not None:
        cfg["train"]["shuffle"] = bool(shuffle)
    return cfg


def build_wordpiece_tokenizer_dir(path, seed=0):
    """BERT-style tokenizer ([CLS]=0 [PAD]=1 [SEP]=2 [UNK]=3, BertNormalizer with clean_text: control characters are
    DELETED, so a word token made only of them receives no sub-token at all -- flair/embeddings.py:3306-3308's case)"""
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, decoders
    from transformers import PreTrainedTokenizerFast
    # hand-built vocabulary (WordPieceTrainer's piece order is not reproducible run to run): specials, characters, their
    # continuation forms, then the corpus words
    chars = "abcdefghijklmnopqrstuvwxyz0123456789.,!()-"
    vocab = ["[CLS]", "[PAD]", "[SEP]", "[UNK]"] + list(chars) + ["##" + c for c in chars] + sorted(set(w.lower() for w in WORDS))
    tok = Tokenizer(models.WordPiece(vocab={w: i for i, w in enumerate(vocab)}, unk_token="[UNK]"))
    tok.normalizer = normalizers.BertNormalizer(clean_text=True, handle_chinese_chars=False, strip_accents=False, lowercase=True)
    tok.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    tok.decoder = decoders.WordPiece()
    os.makedirs(path, exist_ok=True)
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, cls_token="[CLS]", sep_token="[SEP]", pad_token="[PAD]", unk_token="[UNK]",
                                   model_max_length=512)
    fast.save_pretrained(path)
    return fast


def build_bert_dir(path, hidden=128, layers=4, heads=2, inter=256, seed=0):
    """tiny BERT-shaped model directory (word-piece tokenizer + vocab.txt, config.json with model_type bert, random-init weights
    under HF BertModel names): what flair's BertEmbeddings (config 5's mBERT slot) loads"""
    import torch
    from safetensors.torch import save_file
    tok = build_wordpiece_tokenizer_dir(path, seed=seed)
    vocab = sorted(tok.get_vocab().items(), key=lambda kv: kv[1])
    with open(os.path.join(path, "vocab.txt"), "w") as f:
        f.write("\n".join(w for w, _ in vocab) + "\n")
    V = len(vocab)
    cfg = dict(model_type="bert", architectures=["BertModel"], vocab_size=V, hidden_size=hidden, num_hidden_layers=layers,
               num_attention_heads=heads, intermediate_size=inter, max_position_embeddings=512, type_vocab_size=2, pad_token_id=1,
               layer_norm_eps=1e-12, hidden_act="gelu", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f)
    g = torch.Generator().manual_seed(seed)

    def w(*shape, std=0.05):
        return torch.empty(*shape).normal_(0, std, generator=g)

    sd = {"embeddings.word_embeddings.weight": w(V, hidden), "embeddings.position_embeddings.weight": w(512, hidden),
          "embeddings.token_type_embeddings.weight": w(2, hidden), "embeddings.LayerNorm.weight": torch.ones(hidden) + w(hidden),
          "embeddings.LayerNorm.bias": w(hidden)}
    for i in range(layers):
        p = "encoder.layer.%d." % i
        for nm in ("query", "key", "value"):
            sd[p + "attention.self.%s.weight" % nm] = w(hidden, hidden, std=0.08)
            sd[p + "attention.self.%s.bias" % nm] = w(hidden)
        sd[p + "attention.output.dense.weight"] = w(hidden, hidden, std=0.08)
        sd[p + "attention.output.dense.bias"] = w(hidden)
        sd[p + "attention.output.LayerNorm.weight"] = torch.ones(hidden) + w(hidden)
        sd[p + "attention.output.LayerNorm.bias"] = w(hidden)
        sd[p + "intermediate.dense.weight"] = w(inter, hidden, std=0.08)
        sd[p + "intermediate.dense.bias"] = w(inter)
        sd[p + "output.dense.weight"] = w(hidden, inter, std=0.08)
        sd[p + "output.dense.bias"] = w(hidden)
        sd[p + "output.LayerNorm.weight"] = torch.ones(hidden) + w(hidden)
        sd[p + "output.LayerNorm.bias"] = w(hidden)
    save_file(sd, os.path.join(path, "model.safetensors"), metadata={"format": "pt"})
    return path


def kd_golden_batch(g, c):
    """one case of tests/golden/kd_loss.npz as the mirror sees it during distill_mode training: flair Sentences carrying the
    teacher targets the way ModelFinetuner.assign_pretrained_teacher_targets stores them (host arrays trimmed to the sentence),
    a stand-in for the tagger's switches, and the host batch keys FastSequenceTagger._kd_batch reads -> (fake tagger, sentences, hb)"""
    import types
    import numpy as np
    from flair.data import Sentence
    es, lens, tags = g["c%d_es" % c], g["c%d_lens" % c], g["c%d_tags" % c]
    B, n, T = es.shape
    nt = int(g["c%d_n_teachers" % c])
    posterior, crf, att, exact = [bool(x) for x in g["c%d_flags" % c]]
    with_gold, exp_score = [bool(x) for x in g["c%d_with_gold" % c]] if ("c%d_with_gold" % c) in g else (False, False)
    sents = []
    for b in range(B):
        L = int(lens[b])
        s = Sentence(" ".join("w%d" % i for i in range(L)))
        for t in range(nt):
            if crf:
                s.set_teacher_target(g["c%d_t%d_decode" % (c, t)][b, :L])
                if att:
                    s.set_teacher_weights(g["c%d_t%d_path_score" % (c, t)][b])
            if posterior:
                s.set_teacher_posteriors(g["c%d_t%d_fb_score" % (c, t)][b, :L])
            if exact:
                s.set_teacher_posteriors(g["c%d_t%d_pair" % (c, t)][b, :max(L - 1, 0)])
                s.set_teacher_startscores(g["c%d_t%d_start_score" % (c, t)][b])
                s.set_teacher_endscores(g["c%d_t%d_end_score" % (c, t)][b])
        sents.append(s)
    fake = types.SimpleNamespace(tagset_size=T, distill_posterior=posterior, distill_crf=crf, crf_attention=att, distill_exact=exact,
                                 distill_with_gold=with_gold, exp_score=exp_score, distill_emission=False, distill_prob=False,
                                 gold_const=float(g["c%d_gold_const" % c]) if with_gold else 1.0)
    hb = {"row_idx": np.zeros(B * n, np.int32), "tags": tags.astype(np.int32)}
    return fake, sents, hb


def kd_emission_golden_batch(g, c):
    """one case of tests/golden/kd_emission.npz the same way: the sentences carry what ModelFinetuner.
    assign_pretrained_teacher_predictions stores (the teacher's emissions, softmax of them under distill_prob, trimmed to the
    sentence) or, with distill_posterior also on, the teacher's forward-backward scores"""
    import types
    import numpy as np
    from flair.data import Sentence
    es, lens, tags = g["c%d_es" % c], g["c%d_lens" % c], g["c%d_tags" % c]
    B, n, T = es.shape
    prob, posterior = [bool(x) for x in g["c%d_flags" % c]]
    sents = []
    for b in range(B):
        L = int(lens[b])
        s = Sentence(" ".join("w%d" % i for i in range(L)))
        for t in range(int(g["c%d_n_teachers" % c])):
            lg = g["c%d_t%d_logits" % (c, t)][b, :L].astype(np.float64)
            if prob:
                e = np.exp(lg - lg.max(-1, keepdims=True))
                lg = e / e.sum(-1, keepdims=True)
            s.set_teacher_prediction(lg.astype(np.float32))
        if posterior:
            s.set_teacher_posteriors(g["c%d_t0_fb_score" % c][b, :L])
        sents.append(s)
    fake = types.SimpleNamespace(tagset_size=T, distill_posterior=posterior, distill_crf=False, crf_attention=False, distill_exact=False,
                                 distill_with_gold=False, exp_score=False, gold_const=1.0, distill_emission=True, distill_prob=prob)
    hb = {"row_idx": np.zeros(B * n, np.int32), "tags": tags.astype(np.int32)}
    return fake, sents, hb
