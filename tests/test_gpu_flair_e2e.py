"""GPU: the drop-in path end to end -- YAML -> ConfigParser -> FastSequenceTagger -> ModelFinetuner.train on a tiny KB-NER-style
corpus (sentence <EOS> context, B-X context tags, remove_x) -- plus full-size (BASELINE config) invariants of the CRF / encoder."""
import os

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    import tiny_assets
    d = tmp_path_factory.mktemp("e2e")
    tiny_assets.build_model_dir(str(d / "xlmr-tiny"))
    tiny_assets.write_conll_corpus(str(d / "data"), n_train=32, n_dev=8, n_test=8)
    cfg = {
        "ModelFinetuner": {"distill_mode": False, "sentence_level_batch": True},
        "embeddings": {"TransformerWordEmbeddings-0": {"fine_tune": True, "layers": "-1", "model": str(d / "xlmr-tiny"),
                                                        "pooling_operation": "first"}},
        "model": {"FastSequenceTagger": {"crf_attention": False, "dropout": 0.0, "hidden_size": 256, "locked_dropout": 0.0,
                                         "remove_x": True, "sentence_loss": True, "use_cnn": False, "use_crf": True,
                                         "use_rnn": False, "word_dropout": 0.1}},
        "model_name": "tiny_run", "target_dir": str(d / "out"), "targets": "ner", "trainer": "ModelFinetuner",
        "ner": {"Corpus": "ColumnCorpus-TINY", "tag_dictionary": str(d / "tags.pkl"),
                "ColumnCorpus-TINY": {"column_format": {0: "text", 1: "pos", 2: "upos", 3: "ner"}, "comment_symbol": "# id",
                                      "data_folder": str(d / "data"), "tag_to_bioes": "ner"}},
        "train": {"embeddings_storage_mode": "none", "fine_tune_mode": True, "gradient_accumulation_steps": 2,
                  "learning_rate": 2.0e-3, "lr_rate": 50, "max_epochs": 6, "mini_batch_size": 4, "monitor_test": False,
                  "save_finetuned_embedding": True, "select_model_by_macro": True, "train_with_dev": False,
                  "true_reshuffle": False, "use_warmup": False},
    }
    with open(d / "cfg.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    return d


def test_yaml_to_trained_model(workdir):
    import flair
    from flair.config_parser import ConfigParser
    from flair.trainers import ModelFinetuner
    from flair.utils.from_params import Params
    torch.manual_seed(1)
    cp = ConfigParser(Params.from_file(str(workdir / "cfg.yaml")))
    td = cp.tag_dictionary
    assert "S-X" in td.get_items() and td.get_items()[-2:] == ["<START>", "<STOP>"]
    student = cp.create_student()
    names = [n for n, _ in student.named_parameters()]
    assert names[:3] == ["transitions", "linear.weight", "linear.bias"]
    assert any(n.startswith("embeddings.list_embedding_0.model.encoder.layer.0.attention.self.query") for n in names)
    trainer = ModelFinetuner(student, None, cp.corpus, config=cp.config, **cp.config["ModelFinetuner"])
    out = trainer.train(cp.get_target_path, **cp.config["train"])
    hist = out["train_loss_history"]
    assert len(hist) == 6 and hist[-1] < 0.8 * hist[0], hist                 # it learns
    assert len(out["dev_score_history"]) == 6
    base = cp.get_target_path
    for f in ("training.log", "loss.tsv", "final-model.pt", "best-model.pt"):
        assert (base / f).exists(), f
    hf_dir = base / "xlmr-tiny"
    assert (hf_dir / "config.json").exists() and (hf_dir / "model.safetensors").exists() and (hf_dir / "tokenizer.json").exists()
    # prediction file format "token gold pred score" and S-X re-padding of context tokens
    lines = open(base / "ColumnCorpus-TINY-test.tsv").read().strip().split("\n")
    rows = [l.split(" ") for l in lines if l]
    assert all(len(r) == 4 for r in rows)
    assert all(r[2] == "S-X" and float(r[3]) == 1.0 for r in rows if r[1] == "S-X")
    assert all(r[2] != "S-X" for r in rows if r[1] != "S-X")
    # reload round trip: same predictions
    from flair.models import FastSequenceTagger
    again = FastSequenceTagger.load(base / "best-model.pt")   # final_test() left the best model in `student`
    from flair.custom_data_loader import ColumnDataLoader
    dl = ColumnDataLoader(list(cp.corpus.test), 8, sentence_level_batch=True)
    dl.assign_tags("ner", td)
    r1, _ = student.evaluate(dl)
    r2, _ = again.evaluate(dl)
    assert r1.log_line == r2.log_line
    # the saved HF directory is a valid next-stage `embeddings.model`
    from flair.embeddings import TransformerWordEmbeddings
    nxt = TransformerWordEmbeddings(model=str(hf_dir), layers="-1", pooling_operation="first", fine_tune=True)
    w0 = nxt.model.state_dict()["encoder.layer.0.output.dense.weight"]
    w1 = student.engine.hf_state_dict()["encoder.layer.0.output.dense.weight"].cpu()
    assert torch.allclose(w0, w1)
    # posterior (forward-backward marginal) decoding: same API, per-token probabilities in (0, 1]
    student.predict_posterior = True
    try:
        batch = dl[0]
        feats = student.forward(batch)
        labels, _ = student._obtain_labels(feats, batch)
        assert [len(x) for x in labels] == [len(sn) for sn in batch]
        assert all(0.0 < lab.score <= 1.0 for row in labels for lab in row)
    finally:
        student.predict_posterior = False
    # a test sentence longer than one encoder window goes through the sliding-window path (dev/test files are written
    # with max_len=999 by kb/context_process.py:999-1000): same tags for the real tokens as with one big window
    emb = student.embeddings.embeddings[0] if hasattr(student.embeddings, "embeddings") else student.embeddings
    from flair.data import Sentence
    import tiny_assets
    rng = np.random.default_rng(5)
    long_s = Sentence("alice visited berlin <EOS> " + " ".join(str(w) for w in rng.choice(tiny_assets.WORDS, size=150)))
    for t in long_s:
        t.add_tag("ner", "O")
    old = (emb.max_subtokens_sequence_length, emb.stride)
    try:
        student.eval()
        f_one = student.forward([long_s]).clone()
        emb.max_subtokens_sequence_length, emb.stride = 128, 64
        rows, frow, _ = emb.tokenize_sentence(long_s)
        assert len(rows) >= 2 and max(frow) == len(rows) - 1
        f_win = student.forward([long_s])
        assert f_win.shape == f_one.shape
        assert torch.isfinite(f_win).all()
        labels, _ = student._obtain_labels(f_win, [long_s])
        assert len(labels[0]) == len(long_s)
    finally:
        emb.max_subtokens_sequence_length, emb.stride = old


def test_full_size_invariants():
    """BASELINE sizes (XLM-R-large dims, S=512, T=29): size-independent properties instead of an oracle run."""
    from kbner import batch as kb
    from kbner import engine, ops
    T, start, stop, x_idx = 29, 27, 28, 9
    cfg = engine.EncoderConfig(vocab_size=250002, num_hidden_layers=2)      # large width, 2 of the 24 identical layers
    tg = engine.Tagger(cfg, T, start, stop)
    tg.init_random()
    b = kb.synthetic_batch(2, 512, vocab=250002, T=T, x_idx=x_idx, start=start, stop=stop)
    bd = kb.to_device(b)
    loss = tg.forward_loss(bd, backward=True)
    torch.cuda.synchronize()
    assert np.isfinite(float(loss)) and float(loss) > 0                      # logZ >= gold path score
    a = tg.arena
    # emission gradients are (marginals - one-hot): the head bias gradient sums to 0; untouched embedding rows get 0 gradient
    assert abs(float(a.grad("linear.bias").sum())) < 1e-3
    used = torch.unique(torch.from_numpy(b["input_ids"].reshape(-1)).to("cuda"))
    gw = a.grad("emb.word")
    mask = torch.ones(gw.shape[0], dtype=torch.bool, device="cuda")
    mask[used] = False
    assert float(gw[mask].abs().max()) == 0.0 and float(gw[used].abs().max()) > 0.0
    # START row / STOP column of the transitions never receive probability mass
    gt = a.grad("transitions")
    assert float(gt[start, :].abs().max()) == 0.0 and float(gt[:, stop].abs().max()) == 0.0
    # Viterbi path score <= logZ, and decoding is idempotent / deterministic
    em = tg.forward_features(bd)
    lens = bd["lengths"]
    t1, c1 = tg.viterbi(em, lens)
    t2, c2 = tg.viterbi(em, lens)
    assert torch.equal(t1, t2) and torch.equal(c1, c2)
    tags = t1.clamp(min=0).contiguous()
    logz, gold, _ = ops.crf_nll_fwd(em.contiguous(), a.param("transitions"), tags, lens, start, stop)
    assert bool((gold <= logz + 1e-3).all())                                 # best path score is a lower bound of logZ
    # linearity of the GEMM path: Y(2x) - b = 2 (Y(x) - b)
    M, H = 256, 1024
    x = (torch.randn(M, H, device="cuda") * 0.1).to(torch.bfloat16)
    w = a.bf("l0.ffn1.weight")
    y1 = torch.zeros(M, 4096, dtype=torch.bfloat16, device="cuda")
    y2 = torch.zeros_like(y1)
    ops.gemm(0, x, w, M, 4096, H, C=y1)
    ops.gemm(0, (x.float() * 2).to(torch.bfloat16), w, M, 4096, H, C=y2)
    torch.cuda.synchronize()
    assert float((y2.float() - 2 * y1.float()).abs().max()) <= 2e-2 * float(y1.float().abs().max())
    # attention at full size: with V == 1 every output equals the row sum of the softmax = 1 (ragged key mask included);
    # with probability dropout the outputs are unbiased: E[mask / (1-p)] = 1
    B, S, A = 4, 512, 16
    Hh = A * 64
    qkv = (torch.randn(B * S, 3 * Hh, device="cuda") * 0.5).to(torch.bfloat16)
    qkv[:, 2 * Hh:] = 1.0
    am = torch.ones(B, S, device="cuda")
    am[1, 300:] = 0
    mb = ((1 - am) * -10000.0).float().contiguous()
    ctx = torch.zeros(B * S, Hh, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B, A, S, device="cuda")
    ops.attn_fwd(qkv, mb, ctx, lse, B, S, Hh, A)
    torch.cuda.synchronize()
    assert float((ctx.float() - 1.0).abs().max()) < 1e-2
    ctx_d = torch.zeros_like(ctx)
    ops.attn_fwd(qkv, mb, ctx_d, lse, B, S, Hh, A, drop=(2024, ops.drop_thresh(0.1)))
    torch.cuda.synchronize()
    assert abs(float(ctx_d.float().mean()) - 1.0) < 5e-3 and float(ctx_d.float().std()) > 1e-3


def _run_bench(extra_env, launcher, args):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.update(extra_env)
    cmd = [sys.executable] + launcher + [os.path.join(root, "bench.py")] + args
    r = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines            # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_contract_single_gpu():
    """bench.py's CLI / JSON contract on a small configuration (base model, 2 sentences): keys, types, roofline + cpu_baseline"""
    d = _run_bench({}, [], ["--gpus", "1", "--steps", "2", "--warmup", "1", "--model", "base", "--micro-batch", "2", "--accum", "2",
                            "--cpu-sentences", "1"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "sentences/sec" and d["value"] > 0
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["higher_is_better"] is True and d["data"] == "synthetic"
    assert abs(d["value"] - 4 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-2       # value = global batch / step time
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "sample" in cb


def test_bench_two_ranks_one_gpu_functional():
    """the N>1 launch line of the contract (torch.distributed.run, one process per rank) on the single GPU of this box, with
    the gloo escape hatch for the gradient all-reduce: rendezvous, per-rank shards, max-over-ranks timing, rank-0 JSON"""
    d = _run_bench({"KBNER_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"},
                   ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", "29531"],
                   ["--gpus", "2", "--steps", "2", "--warmup", "1", "--model", "base", "--micro-batch", "2", "--accum", "1",
                    "--no-roofline"])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 4 and d["value"] > 0
    assert "cpu_baseline" not in d           # rank 0 at N=1 only


def test_full_size_step_vs_oracle():
    """BASELINE size for real: XLM-R-large dimensions (L24/H1024/A16/F4096, V=250002), two ragged 512-token sentences, one
    forward + backward on the HIP path vs the oracle's fp32 autograd on the host CPU (~1 min)."""
    from kbner import selftest as st
    r = st.check_step(H=1024, A=16, F_=4096, L=24, S=512, V=250002, std=0.02)
    assert r["loss_rel"] < 3e-2, r
    assert r["emissions_rel"] < 5e-2, r
    assert r["grad_min_cos"] > 0.95 and r["grad_worst_rel"] < 0.3, r
    assert r["grad_linear.weight"] < 5e-2 and r["grad_transitions"] < 5e-2, r
    assert r["viterbi_equal"], r


def test_accumulation_fusion_is_the_same_gradient(workdir):
    """ModelFinetuner's fuse_accumulation: the micro-batches of one accumulation group run as one weighted batch == the
    per-micro-batch `loss / accumulate` backward passes summed (finetune_trainer.py:939-957)."""
    from flair.config_parser import ConfigParser
    from flair.custom_data_loader import ColumnDataLoader
    from flair.utils.from_params import Params
    torch.manual_seed(3)
    cp = ConfigParser(Params.from_file(str(workdir / "cfg.yaml")))
    student = cp.create_student()
    student.train()
    student.engine.word_dropout = 0.0                      # the only stochastic site of the tiny config (HF dropout is 0)
    dl = ColumnDataLoader(list(cp.corpus.train), 1, sentence_level_batch=True)     # the YAMLs' mini_batch_size: 1
    dl.assign_tags("ner", cp.tag_dictionary)
    group = [dl[3], dl[11], dl[17], dl[29]]                # different lengths
    accum = len(group)
    g = student.engine.arena.g
    g.zero_()
    l_un = [float(student.forward_backward(b, loss_scale=1.0 / accum)) for b in group]
    g_unfused = g.clone()
    g.zero_()
    sents = [s for b in group for s in b]
    l_f = float(student.forward_backward(sents, loss_scale=1.0, sentence_weights=[1.0 / (accum * len(b)) for b in group for _ in b]))
    torch.cuda.synchronize()
    assert abs(l_f - sum(l_un) / accum) < 2e-2 * abs(l_f)
    a, b = g.double(), g_unfused.double()
    cos = float((a @ b) / (a.norm() * b.norm()))
    assert cos > 0.995, cos                                # bf16 kernels, different padding: not bit-identical
    assert abs(float(a.norm() / b.norm()) - 1.0) < 2e-2
