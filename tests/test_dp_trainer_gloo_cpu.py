"""CPU, world_size 2 over gloo: the data-parallel control flow of ModelFinetuner.train itself, with the device
parts replaced by host stand-ins -- a flat fp32 "arena", an SGD stand-in for FusedAdamW, torch row gather / scatter for the
HIP row kernels -- so that what is tested is the trainer:

  * gradient_accumulation_steps > 1: the overlapped exchange (`GradReducer.bucket_ready`) is armed ONLY on the micro-batch that
    closes an accumulation group (finetune_trainer.py: `hook = ... if flush`); earlier micro-batches accumulate locally;
  * both ranks take the same number of optimiser steps and end with identical parameters, equal to the mean gradient of all
    micro-batches of the step;
  * dev / test evaluation is shared out over the ranks and the tp / fp / fn counters summed: every rank gets the Result that one
    rank scoring every batch gets (the real FastSequenceTagger.evaluate, with labels from a stand-in decoder);
  * files (loss.tsv, best-model.pt) are written by rank 0 only."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


V, H, NPARAM = 40, 8, 3000
EMB_LO = 1000


def _build(folder):
    import tiny_assets
    from flair.data import Label
    from flair.datasets import ColumnCorpus
    from flair.list_data import ListCorpus
    from flair.models import FastSequenceTagger
    corpus = ColumnCorpus(folder, {0: "text", 1: "pos", 2: "upos", 3: "ner"}, tag_to_bioes="ner", comment_symbol="# id")
    td = corpus.make_tag_dictionary("ner")
    lc = ListCorpus([corpus.train], [corpus.dev], [corpus.test], targets=["ColumnCorpus-TINY"])

    class Arena:
        def __init__(self):
            self.p = torch.zeros(NPARAM)
            self.g = torch.zeros(NPARAM)
            self.offsets = {"emb.word": EMB_LO}
            self.shapes = {"emb.word": (V, H)}
            self.emb_flags = torch.zeros(V, dtype=torch.uint8)

        def refresh_shadow(self):
            pass

    class Engine:
        def __init__(self):
            self.arena = Arena()
            self.dynamic_tiles = False
            self._drop_rng = np.random.default_rng(0)

    def word_id(text):
        return sum(ord(c) for c in text) % V

    model = FastSequenceTagger.__new__(FastSequenceTagger)
    torch.nn.Module.__init__(model)
    model.tag_type, model.remove_x, model.tag_dictionary = "ner", True, td
    model.mask = None
    model.engine = Engine()
    model.calls = []          # (sentences in the call, grad_ready armed?)
    model.forward = lambda batch, prediction_mode=False: None
    model._calculate_loss = lambda feats, batch, mask: torch.tensor(float(len(batch)))

    def obtain(feats, batch, get_all_tags=False):
        out = []
        for s in batch:
            labs = []
            for k, t in enumerate(s):
                gold = t.get_tag("ner").value
                # a deterministic "decoder": wrong on some entity tokens, so precision / recall / F1 are not trivially 1
                wrong = gold != "O" and not gold.endswith("-X") and (word_id(t.text) + k) % 3 == 0
                labs.append(Label("O" if wrong else gold, 1.0))
            out.append(labs)
        return out, []

    model._obtain_labels = obtain

    def touched_word_ids(sents):
        return np.unique(np.asarray([word_id(t.text) for s in sents for t in s], np.int64))

    def forward_backward(batch, loss_scale=1.0, sentence_weights=None, grad_ready=None, multi_view=None):
        a = model.engine.arena
        wts = sentence_weights if sentence_weights is not None else [loss_scale / len(batch)] * len(batch)
        for s, w in zip(batch, wts):
            for k, t in enumerate(s):
                wid = word_id(t.text)
                a.g[EMB_LO + wid * H:EMB_LO + (wid + 1) * H] += w * (1.0 + 0.01 * k)     # "embedding" gradient: touched rows only
                a.emb_flags[wid] = 3
            a.g[:EMB_LO] += w * (len(s) * 0.001)                                          # "encoder weight" gradient, 2 buckets
            a.g[EMB_LO + V * H:] += w * 0.5
        model.calls.append((len(batch), grad_ready is not None))
        if grad_ready is not None:
            grad_ready(500, 1000)      # the order backward finishes them: top bucket first
            grad_ready(0, 500)
        return torch.tensor(float(sum(wts)))

    model.touched_word_ids = touched_word_ids
    model.forward_backward = forward_backward
    model.save = lambda path, *a, **k: open(str(path) + ".saved-by-rank", "a").write("%d\n" % int(os.environ["RANK"]))
    model.train = lambda mode=True: model
    model.eval = lambda: model
    return model, lc


class _FakeOpt:
    """stand-in for kbner.engine.FusedAdamW: p -= lr * grad_scale * g, zero g (the trainer only needs step / lr_lambda / t)"""

    def __init__(self, arena, lr=1.0, t_total=None, warmup=0, **kw):
        self.arena, self.lr, self.t, self.t_total, self.warmup = arena, lr, 0, t_total, warmup
        self.steps = []

    def step(self, grad_scale=1.0):
        self.steps.append((self.arena.g * grad_scale).clone())
        self.arena.p -= self.lr * grad_scale * self.arena.g
        self.arena.g.zero_()
        self.t += 1

    def lr_lambda(self):
        return 1.0

    def state_dict(self):
        return {"t": self.t}


class _TorchRows:   # CPU stand-ins for the HIP row kernels
    gather_rows = staticmethod(lambda src, idx: src.index_select(0, idx.long()))
    scatter_rows = staticmethod(lambda rows, idx, dst: dst.index_copy_(0, idx.long(), rows))
    to_bf16 = staticmethod(lambda x: x.to(torch.bfloat16))
    from_bf16 = staticmethod(lambda y, out: out.copy_(y.float()))


def _worker(rank, world, port, folder, base, fuse, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    for p in (ROOT, os.path.join(ROOT, "kb-ner_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    torch.set_num_threads(1)
    import torch.distributed as dist
    from kbner import dp
    import kbner.engine as eng
    from flair.custom_data_loader import ColumnDataLoader
    from flair.trainers import ModelFinetuner
    dp.init_from_env(backend="gloo")
    eng.FusedAdamW = _FakeOpt
    dp._HipRowOps = _TorchRows
    model, corpus = _build(folder)
    trainer = ModelFinetuner(model, None, corpus)
    res = trainer.train(base, learning_rate=1.0, mini_batch_size=2, max_epochs=1, gradient_accumulation_steps=3, shuffle=False,
                        monitor_test=True, fuse_accumulation=fuse, save_final_model=False, sort_data=False,
                        embeddings_storage_mode="none")
    # the same dev set scored by ONE process, unsharded
    dl = ColumnDataLoader(list(corpus.dev_list[0]), 2, False, sort_data=False, model=model)
    dl.assign_tags("ner", model.tag_dictionary)
    whole, whole_loss = model.evaluate(dl, embeddings_storage_mode="none")
    part, part_loss = model.evaluate(dl, embeddings_storage_mode="none", shard=(rank, world))
    gathered = [None] * world
    dist.all_gather_object(gathered, trainer.optimizer.arena.p.clone())
    out.put({"rank": rank, "calls": list(model.calls), "steps": len(trainer.optimizer.steps),
             "step_grads": [g.numpy().copy() for g in trainer.optimizer.steps], "dev_hist": res["dev_score_history"],
             "whole": (whole.main_score, whole.macro_score, whole.log_line, whole.detailed_results, whole_loss),
             "part": (part.main_score, part.macro_score, part.log_line, part.detailed_results, part_loss),
             "same_params": bool(torch.equal(gathered[0], gathered[1]))})
    dp.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fuse", [False, True])
def test_trainer_two_ranks_accumulation_and_sharded_eval(tmp_path, fuse):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tiny_assets
    folder = tiny_assets.write_conll_corpus(str(tmp_path / "c"), n_train=14, n_dev=9, n_test=5, seed=2)
    base = str(tmp_path / "run")
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, folder, base, fuse, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([out.get(timeout=300), out.get(timeout=300)], key=lambda d: d["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # the loader's micro-batches are shared out rank-strided (the tail wrapped so that both ranks run the same count n);
    # accumulate 3 -> ceil(n / 3) optimiser steps per rank, the last group possibly shorter
    n_sent = [sum(k for k, _ in d["calls"]) for d in got]
    assert n_sent[0] == n_sent[1] and n_sent[0] >= 7          # 14 training sentences over 2 ranks
    for d in got:
        armed = [a for _, a in d["calls"]]
        if fuse:
            # one fused call per accumulation group, and the exchange is armed on every one of them
            assert all(armed) and len(armed) == d["steps"], d["calls"]
            assert all(k <= 3 * 2 for k, _ in d["calls"])
        else:
            # armed only on the micro-batch that closes a group: every 3rd one, and the last one of the epoch
            n = len(armed)
            assert n >= 4 and armed == [((i + 1) % 3 == 0) or (i == n - 1) for i in range(n)], d["calls"]
            assert d["steps"] == (n + 2) // 3
        assert d["steps"] >= 2 and d["same_params"]
    assert got[0]["steps"] == got[1]["steps"]
    # the exchanged step gradient is the same tensor on both ranks (the mean over ranks of what each accumulated)
    for g0, g1 in zip(got[0]["step_grads"], got[1]["step_grads"]):
        assert np.allclose(g0, g1, rtol=0, atol=1e-6)
        assert float(np.abs(g0).sum()) > 0
    # sharded evaluation == one process scoring everything; and that is what the trainer's dev history holds on both ranks
    for d in got:
        assert d["part"][:4] == d["whole"][:4], (d["part"], d["whole"])
        assert abs(d["part"][4] - d["whole"][4]) < 1e-9
        assert 0.0 < d["whole"][0] < 1.0
    assert got[0]["dev_hist"] == got[1]["dev_hist"] == [got[0]["whole"][0] * 100]
    # files: one header + one epoch line, written once; best model saved once
    lines = open(os.path.join(base, "loss.tsv")).read().strip().split("\n")
    assert len(lines) == 2 and lines[0].startswith("EPOCH")
    assert open(os.path.join(base, "best-model.pt.saved-by-rank")).read().split() == ["0"]
