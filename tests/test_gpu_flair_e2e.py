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
    cfg = tiny_assets.e2e_config(str(d), word_dropout=0.1, max_epochs=6, n_train=32, n_dev=8, n_test=8)
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
    assert all(0.0 <= x <= 100.0 for x in out["dev_score_history"])      # dataset-level macro average, in percent
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
    # round 6: the HBM-bound kernels of SURVEY section 8d in the same line, timed live (LayerNorm fwd / bwd, AdamW dense + rows, the norms)
    rh = d["roofline_hbm"]
    assert rh["bound"] == "hbm" and rh["unit"] == "GB/s" and rh["peak"] == 8000.0, rh
    for k in ("ln_fwd", "ln_bwd", "adamw_kernel", "grad_sqnorm"):
        kk = rh["kernels"][k]
        assert kk["launches"] >= 1 and kk["achieved"] > 0 and abs(kk["frac"] - kk["achieved"] / 8000.0) < 1e-3, (k, kk)


def test_bench_two_ranks_one_gpu_functional():
    """the N>1 launch line of the contract (torch.distributed.run, one process per rank) on the single GPU of this box, with
    the gloo escape hatch for the gradient all-reduce: rendezvous, per-rank shards, max-over-ranks timing, rank-0 JSON"""
    d = _run_bench({"KBNER_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"},
                   ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", "29531"],
                   ["--gpus", "2", "--steps", "2", "--warmup", "1", "--model", "base", "--micro-batch", "2", "--accum", "1",
                    "--no-roofline"])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 4 and d["value"] > 0
    # the overlapped exchange ran: 12 / 4 = 3 buckets of 4 layers for the base model, the tail measured with HIP events
    assert d["dp_exchange"]["buckets"] == 3 and d["allreduce_ms_exposed"] is not None and d["allreduce_ms_exposed"] >= 0.0
    assert d["dp_exchange"]["emb_mode"] in ("sparse", "dense")
    # round 6: the line explains itself (who ran, with which knobs, what each bucket's collective costs on its own)
    x = d["dp_exchange"]
    assert x["world"] == 2 and x["ranks_seen"] == 2 and x["backend"] == "gloo" and x["bucket_layers"] == 4 and x["exchange_delay"] == 0, x
    assert "diagnostics_error" not in x and len(x["bucket_allreduce_us_isolated"]) == 3 and len(x["bucket_bytes"]) == 3, x
    # ... and the knobs the first multi-GPU run can A/B: 8-layer buckets issued one bucket late give the same loss
    e = _run_bench({"KBNER_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"},
                   ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", "29532"],
                   ["--gpus", "2", "--steps", "2", "--warmup", "1", "--model", "base", "--micro-batch", "2", "--accum", "1",
                    "--no-roofline", "--bucket-layers", "8", "--exchange-delay", "1", "--nccl-max-nchannels", "8"])
    assert e["dp_exchange"]["buckets"] == 2 and e["dp_exchange"]["bucket_layers"] == 8 and e["dp_exchange"]["NCCL_MAX_NCHANNELS"] == "8", e["dp_exchange"]
    assert abs(e["loss_last"] - d["loss_last"]) < 1e-3 * abs(d["loss_last"]), (e["loss_last"], d["loss_last"])


def test_trainer_two_ranks_one_gpu_equals_accumulation(tmp_path):
    """The drop-in trainer itself under data parallelism, with the REAL engine (tests/test_dp_trainer_gloo_cpu.py checks its control
    flow on host stand-ins): two ranks of `ModelFinetuner.train` on one GPU (gloo collectives on device tensors), mini_batch_size 2,
    against ONE process that accumulates the same two micro-batches per step.  Same shuffled batch order (rank-shared RNG), same
    number of optimizer steps, the mean gradient either way (overlapped bucket all-reduce + sparse embedding-row exchange + 1/W in
    AdamW against loss / accumulate), sharded evaluation with summed counters: the final parameters, Adam moments, loss and dev-score
    histories agree.  Lazy embedding rows are forced on in both runs: rows a rank never looked up receive gradients from the other
    rank's batches (the exchange marks them touched) and are caught up inside the update."""
    import subprocess
    import sys
    import tiny_assets
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = tmp_path
    cfg = tiny_assets.e2e_config(str(d), word_dropout=0.0, max_epochs=2, shuffle=True, n_train=32, n_dev=8, n_test=8, accum=1,
                                 mini_batch_size=2, save_finetuned_embedding=False)
    with open(d / "cfg.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    worker = os.path.join(root, "tests", "dp_trainer_gpu_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r1 = subprocess.run([sys.executable, worker, str(d / "cfg.yaml"), str(d / "w1.pt"), "2"], cwd=root, env=env,
                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r1.returncode == 0, r1.stdout.decode()[-3000:]
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                         "127.0.0.1", "--master-port", "29541", worker, str(d / "cfg.yaml"), str(d / "w2.pt"), "1"], cwd=root, env=env,
                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r2.returncode == 0, r2.stdout.decode()[-3000:]
    a, b = torch.load(d / "w1.pt"), torch.load(d / "w2.pt")
    assert a["world"] == 1 and b["world"] == 2 and a["t"] == b["t"] == 2 * 8, (a["t"], b["t"])   # 16 batches of 2 -> 8 steps per epoch
    # (the tiny vocabulary makes the exchange dense -- more than half of the rows are touched per step --, which flags every row;
    #  rows that never receive a gradient keep m = v = 0 and do not move either way)
    assert 0 < a["live_rows"] <= b["live_rows"]
    # the clip norm of the mean gradient, step by step: the first step starts from identical parameters -- equal to fp32 summation
    # order; later steps inherit what Adam makes of that noise (an entry whose gradient is ~0 moves by +-lr whatever its sign)
    na, nb = np.asarray(a["clip_norms"]), np.asarray(b["clip_norms"])
    assert na.shape == nb.shape == (16,) and abs(na[0] - nb[0]) < 1e-5 * na[0], (na[:3], nb[:3])
    assert np.abs(na - nb).max() < 5e-3 * na.max(), (na, nb)
    for k, tol in (("p", 5e-3), ("m", 2e-2)):
        x, y = a[k].double(), b[k].double()
        rel = float((x - y).norm() / x.norm())
        assert rel < tol, (k, rel)
    # the reference's epoch loss is the mean of loss / accumulate over the micro-batches (finetune_trainer.py:1003-1004, 1070): the
    # accumulating twin reports half of what the two ranks report
    la, lb = np.asarray(a["train_loss_history"]), np.asarray(b["train_loss_history"])
    assert la.shape == lb.shape == (2,) and np.abs(2.0 * la - lb).max() < 5e-3 * np.abs(lb).max(), (la, lb)
    assert np.allclose(a["dev_score_history"], b["dev_score_history"], atol=10.0), (a["dev_score_history"], b["dev_score_history"])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs: RCCL cannot put two ranks on one device")
def test_bench_two_ranks_rccl():
    """the real thing whenever the box has two GPUs: one process per GPU over RCCL (backend nccl), overlapped bucketed exchange,
    dynamic GEMM tiles; the loss of the first step must equal the single-process loss of the same two shards"""
    d = _run_bench({"HSA_ENABLE_IPC_MODE_LEGACY": "0"},
                   ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", "29533"],
                   ["--gpus", "2", "--steps", "2", "--warmup", "1", "--micro-batch", "16", "--accum", "1", "--no-roofline"])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["value"] > 0
    assert d["dp_exchange"]["buckets"] == 6 and d["allreduce_ms_exposed"] is not None
    # the first box with two GPUs gives a verdict on the OVERLAPPED path: it must not have fallen back to the blocking
    # all-reduce (bench.py reports a failure of the overlapped exchange as dp_exchange.fallback), and what is left exposed
    # behind the end of backward -- the sparse embedding rows + the ~3 MB remainder, DESIGN.md section 6 -- stays under 5 ms
    assert "fallback" not in d["dp_exchange"], d["dp_exchange"]
    assert d["allreduce_ms_exposed"] < 5.0, d["allreduce_ms_exposed"]
    b = _run_bench({"HSA_ENABLE_IPC_MODE_LEGACY": "0"},
                   ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", "29534"],
                   ["--gpus", "2", "--steps", "2", "--warmup", "1", "--micro-batch", "16", "--accum", "1", "--no-roofline",
                    "--blocking-allreduce", "--static-tiles"])
    assert abs(b["loss_last"] - d["loss_last"]) < 1e-3 * abs(b["loss_last"])     # same mean gradient either way
    assert "cpu_baseline" not in d           # rank 0 at N=1 only


def test_full_size_step_vs_oracle():
    """BASELINE size for real: XLM-R-large dimensions (L24/H1024/A16/F4096, V=250002), two ragged 512-token sentences, one
    forward + backward on the HIP path vs the oracle's fp32 autograd on the host CPU (~1 min)."""
    import selftest as st
    r = st.check_step(H=1024, A=16, F_=4096, L=24, S=512, V=250002, std=0.02, bf16_oracle=(True, "flash"))
    print("full-size step:", {k: v for k, v in r.items() if k != "grad_table_top"})
    # Round 3 measured loss 4.0e-4, emissions 1.22e-2 and a worst gradient of cosine 0.9826 / rel 0.16-0.19 (layer 23's
    # query.weight, growing with depth from 0.013 at layer 0).  Round 4 attributed it with two more oracle passes:
    #   * bf16_points=True rounds to bf16 wherever the HIP path stores bf16 -- and stays at 0.026 from the fp32 oracle: storage
    #     rounding is NOT the cause;
    #   * bf16_points="flash" additionally runs the attention backward the way the kernels do (D = rowdot(dO, bf16 O)) -- and
    #     lands at 0.17 from the fp32 oracle with the same growth over the layers: the cause is the cancellation in
    #     dS = P (dP - D) on nearly parallel value rows, which amplifies the rounding of O (include/kbner.h kbner_attn_bwd).
    # The forward now keeps the residual of O in one byte per element and the HIP gradient is at 0.026 / cosine 0.9997 from the
    # fp32 oracle, which is the distance of the storage-rounding oracle itself.  Thresholds: 3x the observed values.
    assert r["loss_rel"] < 1.2e-3, r
    assert r["emissions_rel"] < 3.7e-2, r
    assert r["grad_min_cos"] > 0.999 and r["grad_worst_rel"] < 0.078, r
    assert r["grad_worst_rel_vs_bf16_oracle"] < 0.09 and r["grad_min_cos_vs_bf16_oracle"] > 0.9985, r
    # the attribution itself: the flash-backward oracle reproduces round 3's distance (within 2x), the storage oracle does not
    assert 0.08 < r["flash_oracle_vs_fp32_worst_rel"] < 0.34 and r["bf16_oracle_vs_fp32_worst_rel"] < 0.06, r


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



# ------------------------------------------------------------------ the reference's own entry-script surface, on live objects
def test_train_surface_live(workdir):
    """tests/golden/train_surface.json (ast of the reference's train.py): every attribute it READS on the student / trainer /
    corpus / config-parser / embedding objects exists on the live mirror objects, and the calls it makes are accepted"""
    import json
    import flair
    from flair.config_parser import ConfigParser
    from flair.custom_data_loader import ColumnDataLoader
    from flair.utils.from_params import Params
    surface = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "train_surface.json")))
    cp = ConfigParser(Params.from_file(str(workdir / "cfg.yaml")), all=False, zero_shot=False, other_shot=False, predict=False,
                      save_embedding=False)
    student = cp.create_student(nocrf=False)
    trainer = getattr(flair.trainers, cp.config["trainer"])(student, None, cp.corpus, config=cp.config,
                                                             **cp.config["ModelFinetuner"], is_test=True)
    live = {"tagger": student, "trainer": trainer, "corpus": cp.corpus, "config_parser": cp,
            "embedding": student.embeddings.embeddings[0], "stacked_embeddings": student.embeddings,
            "dictionary": student.tag_dictionary, "flair_module": flair}
    skip = {("embedding", "ee"), ("embedding", "is_hit_elmo"), ("tagger", "is_mst")}
    missing = []
    for role, obj in live.items():
        for name, uses in surface["attributes"].get(role, {}).items():
            if uses["load"] and not name.startswith("__") and (role, name) not in skip and not hasattr(obj, name):
                missing.append("%s.%s" % (role, name))
    assert not missing, missing
    # the loader / evaluate calls of train.py:153-155 (speed test) and :398-400 (parse)
    test_loader = ColumnDataLoader(list(trainer.corpus.test), 32, use_bert=trainer.use_bert, tokenizer=trainer.bert_tokenizer,
                                   sort_data=False, model=student, sentence_level_batch=True)
    test_loader.assign_tags(student.tag_type, student.tag_dictionary)
    student.eval()
    res, loss = student.evaluate(test_loader, embeddings_storage_mode="none", speed_test=True)
    res, loss = student.evaluate(test_loader, out_path=workdir / "parse.conllu", embeddings_storage_mode="none", prediction_mode=True)
    assert res.detailed_results and isinstance(res.main_score, float)
    assert sum(1 for _ in student.named_parameters()) > 3 and hasattr(student, "named_modules")


# ------------------------------------------------------------------ G5 on the PRODUCT path
def test_obtain_labels_product_path_vs_reference_golden(workdir, golden_dir):
    """FastSequenceTagger._obtain_labels (HIP Viterbi + host re-padding) against the reference-captured obtain_labels.npz in
    BOTH orders: (a) evaluate order -- _calculate_loss ran first and narrowed self.mask to the non-S-X tokens, context is
    re-padded with S-X / 1; (b) speed_test order -- self.mask is the plain length mask, the context is decoded too
    (sequence_tagger_model.py:1193-1210, 2618-2622)"""
    from flair.data import Dictionary, Sentence
    from flair.embeddings import StackedEmbeddings, TransformerWordEmbeddings
    from flair.models import FastSequenceTagger
    g = np.load(os.path.join(golden_dir, "obtain_labels.npz"))
    items = [str(x) for x in g["items"]]
    td = Dictionary(add_unk=False)
    for it in items:
        td.add_item(it)
    emb = TransformerWordEmbeddings(model=str(workdir / "xlmr-tiny"), layers="-1", pooling_operation="first", fine_tune=True)
    tagger = FastSequenceTagger(hidden_size=256, embeddings=StackedEmbeddings([emb]), tag_dictionary=td, tag_type="ner",
                                use_crf=True, use_rnn=False, remove_x=True, dropout=0.0, locked_dropout=0.0, word_dropout=0.0)
    assert (tagger.start_idx, tagger.stop_idx) == (int(g["start"]), int(g["stop"])) and tagger.x_idx == int(g["x_idx"])
    tagger.engine.set_param("transitions", torch.from_numpy(g["trans"]))
    feats, lengths, tags = g["feats"], g["lengths"], g["tags"]
    B, n, T = feats.shape

    class _S:   # what _obtain_labels reads: len(sentence) and the loader's tag-id row
        def __init__(self, k, row):
            self.tokens, self.ner_tags = [None] * k, row

        def __len__(self):
            return len(self.tokens)

    sents = [_S(int(lengths[b]), tags[b]) for b in range(B)]
    ft = torch.from_numpy(feats).cuda()
    length_mask = (torch.arange(n)[None, :] < torch.from_numpy(lengths)[:, None]).float().cuda()
    keep = length_mask.bool().cpu().numpy() & (tags != int(g["x_idx"]))
    for order, mask in (("a", torch.from_numpy(keep.astype(np.float32)).cuda()), ("b", length_mask)):
        tagger.mask = mask
        labels, _ = tagger._obtain_labels(ft, sents)
        for b in range(B):
            got_t = np.asarray([td.get_idx_for_item(l.value) for l in labels[b]], np.int32)
            got_c = np.asarray([l.score for l in labels[b]], np.float32)
            np.testing.assert_array_equal(got_t, g["%s%d_tags" % (order, b)], err_msg="%s %d" % (order, b))
            np.testing.assert_allclose(got_c, g["%s%d_conf" % (order, b)], rtol=2e-6, err_msg="%s %d" % (order, b))


# ------------------------------------------------------------------ G12: the whole drop-in path vs the reference's own run
@pytest.fixture(scope="module")
def g12(tmp_path_factory):
    import json
    import tiny_assets
    gold = os.path.join(os.path.dirname(__file__), "golden")
    e2e = json.load(open(os.path.join(gold, "e2e_train.json"), encoding="utf-8"))
    arrs = np.load(os.path.join(gold, "e2e_train.npz"))
    d = tmp_path_factory.mktemp("g12")
    cfg = tiny_assets.e2e_config(str(d), **e2e["config_kwargs"])
    with open(d / "cfg.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    return d, e2e, arrs


def _student_from(d, arrs, prefix):
    from flair.config_parser import ConfigParser
    from flair.utils.from_params import Params
    cp = ConfigParser(Params.from_file(str(d / "cfg.yaml")))
    student = cp.create_student()
    for k in ("linear.weight", "linear.bias", "transitions"):
        student.engine.set_param(k, torch.from_numpy(arrs[prefix + "/" + k]))
    return cp, student


@pytest.mark.parametrize("lazy_rows", [True, "always"])
def test_training_trajectory_vs_reference_run(g12, lazy_rows):
    """(lazy_rows "always": the same run with FusedAdamW.lazy_rows forced on -- the tiny vocabulary would keep the eager row update --
    and the small-batch routes of round 6 as the trainer picks them: the reference's trajectory must come out either way.)
    ModelFinetuner.train on the HIP engine from the reference run's initial head / transitions, same YAML, no dropout, no
    shuffling: per-micro-batch losses, epoch losses (the reference's loss/accum convention), dev scores (percent) and the final
    transitions against what the reference's own ModelFinetuner.train produced (tests/golden/e2e_train.*).
    Tolerances: bf16 GEMMs/attention vs the reference's fp32 (Adam at lr*lr_rate = 0.1 on the transitions amplifies rounding
    over 80 steps): see the asserts -- 3x what the MI355X run observed."""
    from flair.trainers import ModelFinetuner
    d, e2e, arrs = g12
    cp, student = _student_from(d, arrs, "init")
    steps = []
    fb = student.forward_backward

    def spy(*a, **k):
        out = fb(*a, **k)
        steps.append(out)
        return out

    student.forward_backward = spy
    trainer = ModelFinetuner(student, None, cp.corpus, config=cp.config, **cp.config["ModelFinetuner"])
    trainer.lazy_embedding_rows = lazy_rows
    out = trainer.train(cp.get_target_path, fuse_accumulation=False, **cp.config["train"])
    assert bool(trainer.optimizer.lazy_rows) == (lazy_rows == "always")
    mine = [float(x) for x in steps]
    ref = e2e["step_losses"]
    assert len(mine) == len(ref)
    n_ep = len(ref) // len(e2e["train_loss_history"])
    first = max(abs(a - b) / abs(b) for a, b in zip(mine[:n_ep], ref[:n_ep]))
    hist = max(abs(a - b) / abs(b) for a, b in zip(out["train_loss_history"], e2e["train_loss_history"]))
    print("G12 first-epoch step loss rel", first, "epoch loss rel", hist, "dev", out["dev_score_history"], e2e["dev_score_history"])
    # observed on MI355X (round 2): first-epoch step losses 3.9e-4, epoch losses 2.0e-3, dev scores identical in all 10
    # epochs, transition movement cosine 0.957; thresholds = 3x
    assert first < 1.2e-3, (mine[:n_ep], ref[:n_ep])
    assert hist < 6.1e-3, (out["train_loss_history"], e2e["train_loss_history"])
    assert len(out["dev_score_history"]) == len(e2e["dev_score_history"])
    same = sum(abs(a - b) < 1e-9 for a, b in zip(out["dev_score_history"], e2e["dev_score_history"]))
    assert same >= len(e2e["dev_score_history"]) - 2, (out["dev_score_history"], e2e["dev_score_history"])
    t0, t1 = arrs["init/transitions"], arrs["final/transitions"]
    live = t0 > -1e11
    mv_ref = (t1 - t0)[live]
    mv = (student.transitions.detach().cpu().numpy() - t0)[live]
    cos = float((mv @ mv_ref) / (np.linalg.norm(mv) * np.linalg.norm(mv_ref)))
    print("G12 transition movement cosine", cos)
    assert cos > 0.87, cos


@pytest.mark.parametrize("part", ["dev", "test"])
def test_evaluate_with_reference_weights_vs_reference_lines(g12, part):
    """the reference run's TRAINED weights loaded into the HIP engine, then FastSequenceTagger.evaluate on the same loader: the
    prediction file (token, gold, predicted tag) and the Result against the reference's own evaluate output.  bf16 emissions can
    flip a near-tie, so up to 2 % of the real-token tags may differ; when none does, the Result must be identical."""
    from flair.custom_data_loader import ColumnDataLoader
    d, e2e, arrs = g12
    cp, student = _student_from(d, arrs, "final")
    student.engine.load_hf_state_dict({k[len("final_enc/"):]: torch.from_numpy(arrs[k]) for k in arrs.files if k.startswith("final_enc/")})
    student.eval()
    lst = cp.corpus.dev_list if part == "dev" else cp.corpus.test_list
    dl = ColumnDataLoader(list(lst[0]), 4, False, use_bert=False, sort_data=True, sentence_level_batch=True, model=student)
    dl.assign_tags("ner", cp.tag_dictionary)
    res, loss = student.evaluate(dl, out_path=d / (part + ".tsv"), embeddings_storage_mode="none")
    g = e2e["evaluate"][part]
    mine = open(d / (part + ".tsv"), encoding="utf-8").read().split("\n")
    assert len(mine) == len(g["lines"])
    real = diff = 0
    for a, b in zip(mine, g["lines"]):
        if not b:
            assert a == b
            continue
        fa, fb_ = a.split(" "), b.split(" ")
        assert fa[:2] == fb_[:2], (a, b)                 # token text, gold tag
        if fb_[1] == "S-X":
            assert fa[2:] == fb_[2:], (a, b)             # context: S-X 1 (integer, as the reference prints it)
        else:
            real += 1
            diff += fa[2] != fb_[2]
            if fa[2] == fb_[2]:
                assert abs(float(fa[3]) - float(fb_[3])) < 5e-2, (a, b)
    print("G12 evaluate", part, "real tokens", real, "tag flips", diff, "loss", loss, g["eval_loss"])
    assert diff <= max(1, real // 50), (diff, real)
    assert abs(loss - g["eval_loss"]) < 3e-2 * abs(g["eval_loss"]), (loss, g["eval_loss"])
    if diff == 0:
        assert res.log_line == g["log_line"] and res.detailed_results == g["detailed_results"]
        assert res.main_score == g["main_score"] and res.macro_score == g["macro_score"]


# ------------------------------------------------------------------ checkpoint / resume
@pytest.mark.parametrize("lazy_rows", [True, "always"])
def test_checkpoint_resume_equals_uninterrupted(g12, tmp_path, lazy_rows):
    """(lazy_rows "always": with lazy embedding rows forced on -- the checkpoint then holds a MATERIALIZED table and the restart
    restarts the lazy clock from it.)
    4 epochs straight == 2 epochs + checkpoint.pt + a NEW process-like restart from Model.load_checkpoint for 2 more
    (nn.py:69-139, finetune_trainer.py:1261-1277, trainer.py:582): Adam moments, step count (LR decay position), batch order and
    dropout streams all continue.  Atomics in the embedding backward make runs differ by rounding only."""
    import copy
    from flair.models import FastSequenceTagger
    from flair.trainers import ModelFinetuner
    d, e2e, arrs = g12

    def run(epochs_then=None):
        cp, student = _student_from(d, arrs, "init")
        cfg = copy.deepcopy(cp.config["train"])
        cfg.update(max_epochs=4, checkpoint=True, shuffle=True)
        student.use_word_dropout = student.engine.word_dropout = 0.1     # dropout streams are part of what must continue
        student.engine.seed_dropout(123)
        base = tmp_path / (("straight" if epochs_then is None else "resumed") + str(lazy_rows))
        tr = ModelFinetuner(student, None, cp.corpus, config=cp.config, **cp.config["ModelFinetuner"])
        tr.lazy_embedding_rows = lazy_rows
        if epochs_then is None:
            out = tr.train(base, **cfg)
            return out["train_loss_history"], torch.load(base / "final-model.pt", weights_only=False)
        # interrupted after `epochs_then` epochs: the loop is stopped by max_epochs_without_improvement-free means -- a hook
        cfg1 = dict(cfg)
        hist = []
        orig_save = student.save_checkpoint

        class _Stop(Exception):
            pass

        def save_and_stop(path, *a, **k):
            orig_save(path, *a, **k)
            if a[2] == epochs_then:
                raise KeyboardInterrupt      # the trainer's own early-exit path (finetune_trainer.py:1314-1324)

        student.save_checkpoint = save_and_stop
        out1 = tr.train(base, **cfg1)
        hist += out1["train_loss_history"]
        ck = FastSequenceTagger.load_checkpoint(base / "checkpoint.pt")
        assert ck["epoch"] == epochs_then and ck["optimizer_state_dict"]["t"] > 0
        tr2 = ModelFinetuner.load_from_checkpoint(ck, cp.corpus, config=cp.config, **cp.config["ModelFinetuner"])
        tr2.lazy_embedding_rows = lazy_rows
        out2 = tr2.train(base, **cfg)
        assert bool(tr2.optimizer.lazy_rows) == (lazy_rows == "always")
        hist += out2["train_loss_history"]
        return hist, torch.load(base / "final-model.pt", weights_only=False)

    h_a, p_a = run(None)
    h_b, p_b = run(2)
    print("resume: losses", h_a, h_b)
    assert len(h_a) == len(h_b) == 4
    assert max(abs(a - b) / abs(a) for a, b in zip(h_a, h_b)) < 1.2e-3, (h_a, h_b)   # observed 3.9e-4
    for k in ("transitions", "linear.weight"):
        m = p_a[k] > -1e11
        assert float((p_a[k][m] - p_b[k][m]).abs().max()) < 2e-3 * float(p_a[k][m].abs().max()), k
    for k, v in p_a["encoder_state_dict"].items():
        if k.endswith("key.bias"):
            continue   # true gradient 0 (softmax shift invariance): the value is rounding noise that Adam normalises to +-lr steps
        rel = float((v - p_b["encoder_state_dict"][k]).norm() / (v.norm() + 1e-12))
        # the embedding backward's fp32 atomics make two runs differ by rounding, which Adam's normalisation turns into a few
        # 1e-3 of the (small) distance a bias has moved from 0: observed <= 2.1e-3 (embeddings.LayerNorm.bias), losses 4e-4
        assert rel < 7e-3, (k, rel)


def test_v2_doc_forward_on_device(workdir):
    """train.py --v2doc (`embedding.v2_doc = True`, :223-224) with documents assigned by the trainer (assign_doc_id /
    train_with_doc): every sentence is encoded inside its document window; emissions are finite and shaped like the batch"""
    from flair.config_parser import ConfigParser
    from flair.custom_data_loader import ColumnDataLoader
    from flair.trainers import ModelFinetuner
    from flair.utils.from_params import Params
    cp = ConfigParser(Params.from_file(str(workdir / "cfg.yaml")))
    student = cp.create_student()
    trainer = ModelFinetuner(student, None, cp.corpus, config=cp.config, assign_doc_id=True, train_with_doc=True,
                             **cp.config["ModelFinetuner"])
    sents = list(trainer.corpus.test)
    assert all(hasattr(s, "doc") and s.doc[s.doc_pos] is s for s in sents)
    for emb in student.embeddings.embeddings:
        emb.v2_doc = True
    student.eval()
    dl = ColumnDataLoader(sents, 4, sort_data=False, sentence_level_batch=True, model=student)
    dl.assign_tags(student.tag_type, student.tag_dictionary)
    feats = student.forward(dl[0])
    assert feats.shape[0] == len(dl[0]) and feats.shape[1] == max(len(s) for s in dl[0]) and torch.isfinite(feats).all()
    res, loss = student.evaluate(dl, embeddings_storage_mode="none")
    assert np.isfinite(loss)


# ------------------------------------------------------------------ G14: multi-view (cooperative-learning) training vs the reference's run
def test_multiview_training_vs_reference_run(tmp_path):
    """The shape of the shipped *_doc_joint_multiview_posterior_* YAMLs on the tiny paired corpora: ModelFinetuner pairs the
    *DOC corpus with its source (`orig_sent`), and every micro-batch that carries a second view trains
    (1 - rate) * NLL(context view) + rate * T^2 KL(posterior(context view) || posterior(sentence alone)).
    Against tests/golden/multiview_e2e.* captured from the reference's own trainer (oracle/gen_golden_multiview_e2e.py):
    the pairing, the sequence of NLL / KL values of the first epoch (same weights at its start), the epoch losses."""
    import json
    import tiny_assets
    from flair.config_parser import ConfigParser
    from flair.trainers import ModelFinetuner
    from flair.utils.from_params import Params
    gold = os.path.join(os.path.dirname(__file__), "golden")
    ref = json.load(open(os.path.join(gold, "multiview_e2e.json"), encoding="utf-8"))
    arrs = np.load(os.path.join(gold, "multiview_e2e.npz"))
    cfg = tiny_assets.multiview_config(str(tmp_path), **ref["config_kwargs"])
    with open(tmp_path / "cfg.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    cp = ConfigParser(Params.from_file(str(tmp_path / "cfg.yaml")))
    assert cp.tag_dictionary.get_items() == ref["tag_dictionary"]
    student = cp.create_student()
    assert student.multi_view_training and student.distill_posterior and float(student.temperature) == ref["temperature"]
    for k in ("linear.weight", "linear.bias", "transitions"):
        student.engine.set_param(k, torch.from_numpy(arrs["init/" + k]))
    trainer = ModelFinetuner(student, None, cp.corpus, config=cp.config, **cp.config["ModelFinetuner"])
    # the pairing of finetune_trainer.py:316-344
    for name, ci in trainer.corpus2id.items():
        for part in ("train_list", "dev_list", "test_list"):
            got = [s.orig_sent.to_tokenized_string() if hasattr(s, "orig_sent") else None for s in getattr(cp.corpus, part)[ci]]
            assert got == ref["pairing"]["%s/%s" % (name, part)], (name, part)
    calls = []
    fb = student.forward_backward

    def spy(data_points, *a, **k):
        out = fb(data_points, *a, **k)
        nll, kd = student.last_loss_parts
        mv = kd is not None
        rate = ref["multi_view_rate"]
        calls.append(["nll", float(nll) / ((1.0 - rate) if mv else 1.0), [s.to_tokenized_string() for s in data_points]])
        if mv:
            calls.append(["kl", float(kd) / rate, [s.to_tokenized_string() for s in data_points]])
        return out

    student.forward_backward = spy
    out = trainer.train(cp.get_target_path, fuse_accumulation=False, **cp.config["train"])
    rc = ref["calls"]
    assert [c[0] for c in calls] == [c[0] for c in rc], "the same micro-batches must carry a second view"
    assert [c[2] for c in calls] == [c[2] for c in rc], "same batches, same order"
    n_ep = len(rc) // len(ref["train_loss_history"])
    worst_nll = max(abs(a[1] - b[1]) / abs(b[1]) for a, b in zip(calls[:n_ep], rc[:n_ep]) if a[0] == "nll")
    kl = [(a[1], b[1]) for a, b in zip(calls[:n_ep], rc[:n_ep]) if a[0] == "kl"]
    assert len(kl) >= 3
    worst_kl = max(abs(a - b) / max(abs(b), 1e-4) for a, b in kl)
    hist = max(abs(a - b) / abs(b) for a, b in zip(out["train_loss_history"], ref["train_loss_history"]))
    print("G14 first-epoch NLL rel", worst_nll, "KL rel", worst_kl, "epoch loss rel", hist, "kl pairs", kl[:4],
          "dev", out["dev_score_history"], ref["dev_score_history"])
    # observed on MI355X (round 2): NLL 1.5e-3, KL 2.8e-2 (a second-order small quantity, 5e-4 .. 4e-3 here, computed from bf16
    # emissions of two different batches), epoch losses 1.6e-3, dev scores identical in all epochs; thresholds = 3x
    assert worst_nll < 4.5e-3, (calls[:n_ep], rc[:n_ep])
    assert worst_kl < 8.4e-2, kl
    assert hist < 4.8e-3, (out["train_loss_history"], ref["train_loss_history"])
    # dev scores of 8 tiny sentences: one near-tie tag flip (bf16 emissions, atomics order) moves a score by several points, and it
    # did in one of three runs on the second epoch -- the first epoch (12 optimizer steps from identical weights) must agree
    assert abs(out["dev_score_history"][0] - ref["dev_score_history"][0]) < 1e-9, (out["dev_score_history"], ref["dev_score_history"])
    assert len(out["dev_score_history"]) == len(ref["dev_score_history"])

    # the accumulation-group fusion of ModelFinetuner.train carries the second view too: one weighted batch for the group's
    # context views + one for its bare sentences == the per-micro-batch passes summed
    loader = ColumnDataLoaderFor(trainer, cp)
    group = [b for b in loader if student.multi_view_plan(b)][:2] + [b for b in loader if not student.multi_view_plan(b)][:1]
    assert len(group) == 3
    student.forward_backward = fb
    student.train()
    g = student.engine.arena.g
    g.zero_()
    tot_un = 0.0
    for b in group:
        wts, mv = trainer._group_weights([b], ref["multi_view_rate"])
        tot_un += float(student.forward_backward(b, loss_scale=1.0 / len(group), sentence_weights=wts, multi_view=mv)) / len(group)
    g_un = g.clone()
    g.zero_()
    wts, mv = trainer._group_weights(group, ref["multi_view_rate"])
    tot_f = float(student.forward_backward([s for b in group for s in b], loss_scale=1.0, sentence_weights=wts, multi_view=mv))
    torch.cuda.synchronize()
    assert abs(tot_f - tot_un) < 2e-2 * abs(tot_un), (tot_f, tot_un)
    a, b = g.double(), g_un.double()
    cos = float((a @ b) / (a.norm() * b.norm()))
    assert cos > 0.995 and abs(float(a.norm() / b.norm()) - 1.0) < 2e-2, (cos, float(a.norm() / b.norm()))


# ------------------------------------------------------------------ G15: teacher-student knowledge distillation vs the reference's run
@pytest.mark.parametrize("name", ["posterior_crf_att", "exact"])
def test_kd_training_vs_reference_run(tmp_path, name):
    """`ModelFinetuner: {distill_mode: true}` end to end, configured by the reference's keys (tiny_assets.kd_config): the teacher
    is loaded from its own YAML + best-model.pt through ConfigParser.create_teachers_list, labels the training set once
    (assign_pretrained_teacher_targets: n-best paths + weights + forward-backward scores, or the pairwise posteriors of
    distill_exact), and the student trains on interpolation * KD + (1 - interpolation) * NLL.
    Against tests/golden/kd_e2e.* captured from the reference's own trainer with the same frozen teacher
    (oracle/gen_golden_kd_e2e.py): the per-sentence teacher targets, every loss value of the first epoch (same weights at its
    start), the epoch losses."""
    import json
    import tiny_assets
    from flair.config_parser import ConfigParser
    from flair.trainers import ModelFinetuner
    from flair.utils.from_params import Params
    gold = os.path.join(os.path.dirname(__file__), "golden")
    ref = json.load(open(os.path.join(gold, "kd_e2e.json"), encoding="utf-8"))[name]
    arrs = np.load(os.path.join(gold, "kd_e2e.npz"))
    kw = ref["config_kwargs"]
    cfg, tcfg = tiny_assets.kd_config(str(tmp_path), **kw)
    with open(tmp_path / "cfg.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    with open(tmp_path / "teacher.yaml", "w") as f:
        yaml.safe_dump(tcfg, f)
    cp = ConfigParser(Params.from_file(str(tmp_path / "cfg.yaml")))
    assert cp.tag_dictionary.get_items() == ref["tag_dictionary"]
    # the frozen teacher of the reference run: tiny pre-trained encoder + the recorded head, written as ITS best-model.pt
    teacher = cp.create_model(Params.from_file(str(tmp_path / "teacher.yaml")))
    for k in ("linear.weight", "linear.bias", "transitions"):
        teacher.engine.set_param(k, torch.from_numpy(arrs["%s/teacher/%s" % (name, k)]))
    tdir = os.path.join(tcfg["target_dir"], tcfg["model_name"])
    os.makedirs(tdir, exist_ok=True)
    teacher.save(os.path.join(tdir, "best-model.pt"))
    del teacher
    teachers = cp.create_teachers_list()
    assert len(teachers) == 1 and teachers[0].targets == {"ColumnCorpus-TINY"}
    np.testing.assert_array_equal(teachers[0].transitions.cpu().numpy(), arrs[name + "/teacher/transitions"])
    student = cp.create_student()
    assert (student.distill_posterior, student.distill_crf, student.crf_attention, student.distill_exact) == \
        (kw["posterior"], kw["crf"], kw["attention"], kw["exact"])
    for k in ("linear.weight", "linear.bias", "transitions"):
        student.engine.set_param(k, torch.from_numpy(arrs["%s/init/%s" % (name, k)]))
    trainer = ModelFinetuner(student, teachers, cp.corpus, config=cp.config, professors=[], **cp.config["ModelFinetuner"])
    assert trainer.distill_mode and trainer.interpolation == kw["interpolation"]
    calls = []
    fb = student.forward_backward

    def spy(data_points, *a, **k):
        out = fb(data_points, *a, **k)
        calls.append([float(out), float(k["distill_interpolation"]), [s.to_tokenized_string() for s in data_points]])
        return out

    student.forward_backward = spy
    out = trainer.train(cp.get_target_path, fuse_accumulation=False, **cp.config["train"])
    assert trainer.teachers == []           # released after labelling (finetune_trainer.py:634-636)
    # ---- the targets the teacher left on the sentences
    sents = list(cp.corpus.train_list[0])
    assert [s.to_tokenized_string() for s in sents] == [r["text"] for r in ref["sentences"]]
    tok = agree = 0
    worst = {"weights": 0.0, "fb_score": 0.0, "pair": 0.0, "start_score": 0.0, "end_score": 0.0}
    for i, s in enumerate(sents):
        pre = "%s/s%d/" % (name, i)
        if kw["crf"]:
            want = arrs[pre + "target"]
            got = s.get_teacher_target()
            assert got.shape == want.shape
            tok += want.size
            agree += int((got == want).sum())
            if kw["attention"]:
                worst["weights"] = max(worst["weights"], float(np.abs(s.get_teacher_weights() - arrs[pre + "weights"]).max()))
        if kw["posterior"]:
            want, got = arrs[pre + "fb_score"], s._teacher_posteriors[0]
            fin = want > -1e10
            assert got.shape == want.shape and (got[~fin] < -1e10).all()
            worst["fb_score"] = max(worst["fb_score"], float(np.abs(got[fin] - want[fin]).max() / max(1.0, np.abs(want[fin]).max())))
        if kw["exact"]:
            want, got = arrs[pre + "pair"], s._teacher_posteriors[0]
            assert got.shape == want.shape
            worst["pair"] = max(worst["pair"], float(np.abs(got - want).max(initial=0.0)))
            for nm, g2 in (("start_score", s._teacher_startscores[0]), ("end_score", s._teacher_endscores[0])):
                want = arrs[pre + nm]
                fin = want > -1e10
                assert (g2[~fin] < -1e10).all()
                worst[nm] = max(worst[nm], float(np.abs(g2[fin] - want[fin]).max() / max(1.0, np.abs(want[fin]).max())))
    rc = ref["calls"]
    assert [c[2] for c in calls] == [c[2] for c in rc], "same batches, same order"
    assert all(abs(a[1] - b[1]) < 1e-12 for a, b in zip(calls, rc)), "same interpolation"
    n_ep = len(rc) // len(ref["train_loss_history"])
    first = max(abs(a[0] - b[0]) / abs(b[0]) for a, b in zip(calls[:n_ep], rc[:n_ep]))
    hist = max(abs(a - b) / abs(b) for a, b in zip(out["train_loss_history"], ref["train_loss_history"]))
    print("G15", name, "n-best agreement %d/%d" % (agree, tok), "targets", worst, "first-epoch loss rel", first, "epoch loss rel", hist,
          "dev", out["dev_score_history"], ref["dev_score_history"])
    if kw["crf"]:
        # the decoder itself is bit-exact on identical inputs (tests/golden/viterbi_nbest.npz, kd_loss.npz); here its input is the
        # teacher's emissions through the bf16 HIP encoder vs the reference's fp32 one, and near-tied candidates swap ranks
        assert agree >= 0.97 * tok, (agree, tok)
    # observed on MI355X (round 3; bf16 HIP encoder vs the reference's fp32 one): n-best agreement 895/906, path weights 2.0e-2,
    # fb scores 3.1e-3 (relative to the largest score), pair posteriors 9.2e-3, start / end scores 3.0e-3; first-epoch losses
    # 4.4e-3 / 6.1e-4, epoch losses 8.6e-4 / 2.4e-4 (posterior+crf+attention / exact); thresholds = 3x
    assert worst["weights"] < 6e-2 and worst["fb_score"] < 1e-2 and worst["pair"] < 2.8e-2, worst
    assert worst["start_score"] < 9e-3 and worst["end_score"] < 9e-3, worst
    assert first < 1.4e-2, (calls[:n_ep], rc[:n_ep])
    assert hist < 2.6e-3, (out["train_loss_history"], ref["train_loss_history"])
    assert len(out["dev_score_history"]) == len(ref["dev_score_history"])


@pytest.mark.parametrize("prob", [False, True])
def test_kd_emission_training(tmp_path, prob):
    """`distill_mode` with `distill_emission` (+ `distill_prob`) through the trainer: the teacher labels the training set with its
    emissions (assign_pretrained_teacher_predictions: teacher.forward, softmax under distill_prob), and the student's first loss is
    interpolation * T^2 KL(teacher || student emissions) / B + (1 - interpolation) * NLL -- checked against the oracle restatement
    (pinned to the reference's own loss by tests/golden/kd_emission.npz) evaluated on the student's emissions of that batch"""
    import tiny_assets
    from flair.config_parser import ConfigParser
    from flair.trainers import ModelFinetuner
    from flair.utils.from_params import Params
    from oracle import kd as okd
    cfg, tcfg = tiny_assets.kd_config(str(tmp_path), posterior=False, crf=False, attention=False, exact=False, temperature=2.0,
                                      interpolation=0.6, max_epochs=3)
    cfg["model"]["FastSequenceTagger"].update(distill_emission=True, distill_prob=prob)
    with open(tmp_path / "cfg.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    with open(tmp_path / "teacher.yaml", "w") as f:
        yaml.safe_dump(tcfg, f)
    cp = ConfigParser(Params.from_file(str(tmp_path / "cfg.yaml")))
    teacher = cp.create_model(Params.from_file(str(tmp_path / "teacher.yaml")))
    g = torch.Generator().manual_seed(11)
    T = len(cp.tag_dictionary)
    teacher.engine.set_param("linear.weight", torch.randn(T, teacher.engine.cfg.hidden_size, generator=g) * 0.3)
    tdir = os.path.join(tcfg["target_dir"], tcfg["model_name"])
    os.makedirs(tdir, exist_ok=True)
    teacher.save(os.path.join(tdir, "best-model.pt"))
    sents = list(cp.corpus.train_list[0])
    want = {}
    for s in sents:                       # the teacher's emissions, one sentence at a time
        em = teacher.forward([s])[0, :len(s)].float().cpu()
        want[s.to_tokenized_string()] = (torch.softmax(em, -1) if prob else em).numpy()
    del teacher
    teachers = cp.create_teachers_list()
    student = cp.create_student()
    assert student.distill_emission and student.distill_prob == prob
    trainer = ModelFinetuner(student, teachers, cp.corpus, config=cp.config, professors=[], **cp.config["ModelFinetuner"])
    calls = []
    fb = student.forward_backward

    def spy(data_points, *a, **k):
        ip = float(k["distill_interpolation"])
        with torch.no_grad():
            student.eval()
            es = student.forward(data_points).float().cpu()
            student.train()
        before = (ip, es, list(data_points), student.transitions.float().cpu().clone())
        out = fb(data_points, *a, **k)
        calls.append((float(out),) + before)
        return out

    student.forward_backward = spy
    out = trainer.train(cp.get_target_path, fuse_accumulation=False, **cp.config["train"])
    assert trainer.teachers == []
    worst = 0.0
    for s in sents:
        got = s.get_teacher_prediction()
        ref = want[s.to_tokenized_string()]
        assert got.shape == ref.shape
        worst = max(worst, float(np.abs(got - ref).max() / max(1.0, np.abs(ref).max())))
    # (batched vs one-at-a-time teacher forward: the same kernels on differently padded batches)
    assert worst < 2e-2, worst
    # the first micro-batch of the run, before any update: loss == the oracle's on the student's own emissions
    loss0, ip, es, batch, trans0 = calls[0]
    assert ip == 0.6
    B, n, _ = es.shape
    lens = np.asarray([len(s) for s in batch])
    teach = np.zeros((B, n, T), np.float32)
    tags = np.zeros((B, n), np.int64)
    for b, s in enumerate(batch):
        teach[b, :len(s)] = s.get_teacher_prediction()
        tags[b, :len(s)] = [student.tag_dictionary.get_idx_for_item(t.get_tag("ner").value) for t in s]
    x_idx = student.tag_dictionary.get_idx_for_item("S-X")
    ref = okd.kd_loss(es, trans0, lens, tags, student.start_idx, student.stop_idx, x_idx, 2.0, ip,
                      emission=torch.from_numpy(teach), emission_is_prob=prob)
    print("emission KD first loss", loss0, float(ref), "teacher prediction worst rel", worst, "history", out["train_loss_history"])
    assert abs(loss0 - float(ref)) <= 5e-3 * abs(float(ref)), (loss0, float(ref))
    assert out["train_loss_history"][-1] < out["train_loss_history"][0]


def ColumnDataLoaderFor(trainer, cp):
    """the training loader ModelFinetuner.train builds (same arguments)"""
    from flair.custom_data_loader import ColumnDataLoader
    data = [s for ds in cp.corpus.train_list for s in ds]
    dl = ColumnDataLoader(data, cp.config["train"]["mini_batch_size"], False, use_bert=False, model=trainer.model,
                          sentence_level_batch=trainer.sentence_level_batch, sort_data=True)
    dl.assign_tags(trainer.model.tag_type, trainer.model.tag_dictionary)
    return dl


def test_train_py_command_line(tmp_path):
    """this repo's train.py as a user runs it (subprocesses, the reference's flags): fine-tune, --test, --parse of a folder of
    CoNLL files in file order with --predict_posterior, --test_speed"""
    import subprocess
    import sys
    import tiny_assets
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = tiny_assets.e2e_config(str(tmp_path), word_dropout=0.1, max_epochs=2, n_train=12, n_dev=4, n_test=5)
    with open(tmp_path / "cfg.yaml", "w") as f:
        yaml.safe_dump(cfg, f)

    def run(*extra):
        r = subprocess.run([sys.executable, os.path.join(root, "train.py"), "--config", str(tmp_path / "cfg.yaml")] + list(extra),
                           cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
        return r.stdout.decode(errors="replace") + r.stderr.decode(errors="replace")

    run()
    base = tmp_path / "out" / "tiny_run"
    assert (base / "best-model.pt").exists() and (base / "final-model.pt").exists() and (base / "loss.tsv").exists()
    out = run("--test")
    assert "Testing using best model" in out and (base / "ColumnCorpus-TINY-test.tsv").exists()
    # --parse: a folder with a train.txt in the 4-column KB-NER format, kept in file order, marginal decoding
    tiny_assets.write_conll_corpus(str(tmp_path / "new"), n_train=7, n_dev=1, n_test=1, seed=11)
    out = run("--parse", "--target_dir", str(tmp_path / "new"), "--num_columns", "4", "--comment_symbol", "# id", "--keep_order",
              "--predict_posterior", "--output_dir", str(tmp_path / "parsed"))
    pred = (tmp_path / "parsed" / "new.conllu").read_text().split("\n")
    src_tokens = [l.split(" ")[0] for l in (tmp_path / "new" / "train.txt").read_text().split("\n") if l and not l.startswith("# id")]
    assert [l.split(" ")[0] for l in pred if l] == src_tokens          # every token, in file order
    assert all(len(l.split(" ")) == 4 for l in pred if l)
    out = run("--test_speed")
    assert any(l.strip().replace(".", "", 1).isdigit() for l in out.split("\n"))      # the reference prints the bare rate


def test_train_py_distill_mode_command_line(tmp_path):
    """the KD path as a user runs it: train a teacher with this repo's train.py, then train a student from a YAML that names the
    teacher's YAML (`ModelFinetuner: {distill_mode: true}`, `is_teacher_list`, `ner.teachers`) -- train.py builds the teacher from
    its own config + best-model.pt, labels the training set and trains on interpolation * KD + (1 - interpolation) * NLL;
    --test of the student afterwards builds no teachers"""
    import subprocess
    import sys
    import tiny_assets
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg, tcfg = tiny_assets.kd_config(str(tmp_path), max_epochs=2, posterior=True, crf=True, attention=True, exact=False, best_k=3)
    tcfg["train"]["max_epochs"] = 3
    with open(tmp_path / "cfg.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    with open(tmp_path / "teacher.yaml", "w") as f:
        yaml.safe_dump(tcfg, f)

    def run(config, *extra):
        r = subprocess.run([sys.executable, os.path.join(root, "train.py"), "--config", str(tmp_path / config)] + list(extra),
                           cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
        return r.stdout.decode(errors="replace") + r.stderr.decode(errors="replace")

    run("teacher.yaml")
    assert (tmp_path / "out" / "tiny_teacher" / "best-model.pt").exists()
    out = run("cfg.yaml")
    base = tmp_path / "out" / "tiny_kd_run"
    assert "Distilling sentences as targets" in out and "Distilled 18 sentences" in out and "distill_mode: interpolation" in out
    assert (base / "best-model.pt").exists() and (base / "loss.tsv").exists()
    rows = [l.split("\t") for l in (base / "loss.tsv").read_text().strip().split("\n")[1:]]
    assert len(rows) == 2 and all(float(r[3]) > 0 for r in rows) and float(rows[1][3]) < float(rows[0][3])     # the KD loss goes down
    out = run("cfg.yaml", "--test")
    assert "Distilling" not in out and "Testing using best model" in out


def test_multiview_and_kd_trainers_with_lazy_rows(tmp_path, monkeypatch):
    """The multi-view run (two encoder passes per step: the context view looks its own sub-tokens up) and the teacher-student run,
    both pinned to the reference's own trainer, once more with FusedAdamW.lazy_rows forced on for every ModelFinetuner: same
    assertions against the same goldens."""
    from flair.trainers import ModelFinetuner
    monkeypatch.setattr(ModelFinetuner, "lazy_embedding_rows", "always", raising=False)
    (tmp_path / "mv").mkdir()
    (tmp_path / "kd").mkdir()
    test_multiview_training_vs_reference_run(tmp_path / "mv")
    test_kd_training_vs_reference_run(tmp_path / "kd", "posterior_crf_att")


@pytest.mark.parametrize("flags", [dict(distill_exact=True, distill_posterior=False), dict(calculate_l2_loss=True),
                                   dict(calculate_l2_loss=True, l2_loss_only=True)])
def test_multiview_other_branches_train(tmp_path, flags):
    """the multi-view YAML shape with the OTHER branches of _calculate_multi_view_loss switched on (distill_exact; calculate_l2_loss on
    top of the posterior term; l2_loss_only): the trainer runs, the second-view term is positive and finite in every micro-batch that
    carries one, and the epoch loss goes down.  (The terms themselves are pinned to the reference in tests/test_gpu_kernels.py.)"""
    import tiny_assets
    from flair.config_parser import ConfigParser
    from flair.trainers import ModelFinetuner
    from flair.utils.from_params import Params
    cfg = tiny_assets.multiview_config(str(tmp_path), max_epochs=3, accum=2, mini_batch_size=2, temperature=2.0)
    cfg["model"]["FastSequenceTagger"].update(flags)
    with open(tmp_path / "cfg.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    cp = ConfigParser(Params.from_file(str(tmp_path / "cfg.yaml")))
    student = cp.create_student()
    trainer = ModelFinetuner(student, None, cp.corpus, config=cp.config, **cp.config["ModelFinetuner"])
    parts = []
    fb = student.forward_backward

    def spy(data_points, *a, **k):
        out = fb(data_points, *a, **k)
        nll, kd = student.last_loss_parts
        if kd is not None:
            parts.append(float(kd))
        return out

    student.forward_backward = spy
    out = trainer.train(cp.get_target_path, **cp.config["train"])
    assert len(parts) >= 3 and all(np.isfinite(p) and p >= 0 for p in parts) and max(parts) > 0
    h = out["train_loss_history"]
    assert all(np.isfinite(x) for x in h) and h[-1] < h[0], h


@pytest.mark.parametrize("sentence_loss", [True, False])
def test_softmax_student_through_the_yaml_path(tmp_path, sentence_loss):
    """use_crf: false in the YAML (the reference's softmax student, sequence_tagger_model.py:2523-2539 / :1177-1180): ConfigParser ->
    FastSequenceTagger without a `transitions` parameter -> forward_loss == the oracle's token-level cross entropy of the tagger's
    own emissions (remove_x narrowing, / B or / kept tokens) -> ModelFinetuner.train learns -> evaluate decodes EVERY token by arg-max
    (no S-X re-padding on this branch) -> save / load round trip."""
    import tiny_assets
    from flair.config_parser import ConfigParser
    from flair.custom_data_loader import ColumnDataLoader
    from flair.models import FastSequenceTagger
    from flair.trainers import ModelFinetuner
    from flair.utils.from_params import Params
    from oracle import crf as ocrf
    cfg = tiny_assets.e2e_config(str(tmp_path), word_dropout=0.0, max_epochs=4, n_train=32, n_dev=8, n_test=8)
    cfg["model"]["FastSequenceTagger"]["use_crf"] = False
    cfg["model"]["FastSequenceTagger"]["sentence_loss"] = sentence_loss
    cfg["train"]["fuse_accumulation"] = True     # (the trainer itself must leave the group unfused without sentence_loss)
    with open(tmp_path / "cfg.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    torch.manual_seed(5)
    cp = ConfigParser(Params.from_file(str(tmp_path / "cfg.yaml")))
    td = cp.tag_dictionary
    student = cp.create_student()
    assert student.use_crf is False
    names = [n for n, _ in student.named_parameters()]
    assert "transitions" not in names and names[:2] == ["linear.weight", "linear.bias"]
    dl = ColumnDataLoader(list(cp.corpus.train), 4, sentence_level_batch=True)
    dl.assign_tags("ner", td)
    batch = dl[2]
    feats = student.forward(batch).float().cpu().numpy()
    lengths = np.asarray([len(s) for s in batch])
    tags = np.zeros(feats.shape[:2], np.int64)
    for b, s in enumerate(batch):
        tags[b, :len(s)] = [td.get_idx_for_item(t.get_tag("ner").value) for t in s]
    want, _ = ocrf.softmax_calculate_loss(feats, tags, lengths, x_idx=td.get_idx_for_item("S-X"), sentence_loss=sentence_loss)
    got = float(student.forward_loss(batch))
    assert abs(got - want) <= 2e-3 * abs(want), (got, want)       # (two forward passes of a bf16 encoder: same weights, same batch)
    trainer = ModelFinetuner(student, None, cp.corpus, config=cp.config, **cp.config["ModelFinetuner"])
    out = trainer.train(cp.get_target_path, **cp.config["train"])
    hist = out["train_loss_history"]
    assert len(hist) == 4 and hist[-1] < 0.8 * hist[0], hist
    lines = [l.split(" ") for l in open(cp.get_target_path / "ColumnCorpus-TINY-test.tsv").read().strip().split("\n") if l]
    assert all(len(r) == 4 and 0.0 < float(r[3]) <= 1.0 for r in lines)
    student.save(tmp_path / "softmax-student.pt")
    again = FastSequenceTagger.load(tmp_path / "softmax-student.pt")
    assert again.use_crf is False
    dt = ColumnDataLoader(list(cp.corpus.test), 8, sentence_level_batch=True)
    dt.assign_tags("ner", td)
    r1, l1 = student.evaluate(dt)
    r2, l2 = again.evaluate(dt)
    assert r1.main_score == r2.main_score and abs(l1 - l2) <= 1e-6 * max(1.0, abs(l1))
