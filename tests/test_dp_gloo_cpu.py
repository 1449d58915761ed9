"""CPU, world_size 2 over gloo: the data-parallel host logic (kbner.dp) -- shard coverage, the sum-all-reduce + 1/W contract
that FusedAdamW's grad_scale completes, global step counting -- and that 2-rank gradient averaging reproduces the
single-process gradient of the doubled batch (checked on the oracle's fp32 tagger so it runs without a GPU)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tiny_problem():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
    from kbner import batch as kb
    from oracle import encoder as oenc
    cfg = oenc.EncoderConfig(vocab_size=97, hidden_size=32, num_hidden_layers=1, num_attention_heads=2, intermediate_size=64,
                             max_position_embeddings=40)
    params = oenc.init_params(cfg, seed=3, std=0.2)
    g = torch.Generator().manual_seed(5)
    T, start, stop, x_idx = 8, 6, 7, 5
    params["linear.weight"] = torch.randn(T, 32, generator=g) * 0.3
    params["linear.bias"] = torch.zeros(T)
    tr = torch.randn(T, T, generator=g)
    tr[start, :] = -1e12
    tr[:, stop] = -1e12
    params["transitions"] = tr
    ids, am, first, tags, lengths = kb.synthetic_sentences(4, 32, vocab=97, T=T, x_idx=x_idx, start=start, stop=stop, n_real=4, seed=9)
    return cfg, params, (ids, am, first, tags, lengths), (start, stop, x_idx)


def _grads(cfg, params, data, tagsinfo, rows):
    from oracle import train_step as ots
    ids, am, first, tags, lengths = data
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    b = dict(input_ids=torch.from_numpy(ids[rows]), attention_mask=torch.from_numpy(am[rows]),
             first_idx=torch.from_numpy(first[rows]), tags=torch.from_numpy(tags[rows]), lengths=torch.from_numpy(lengths[rows]))
    loss, _ = ots.tagger_forward_loss(p, cfg, b, *tagsinfo)
    loss.backward()
    names = sorted(p)
    return torch.cat([(p[n].grad if p[n].grad is not None else torch.zeros_like(p[n])).flatten() for n in names]), float(loss)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
    torch.set_num_threads(1)
    from kbner import dp
    dp.init_from_env(backend="gloo")
    assert dp.world_size() == world and dp.rank() == rank
    # 1. shard coverage: every micro-batch index is run, every rank runs the same count
    mine = dp.shard_indices(7)
    assert len(mine) == 4 and mine == [(rank + i * world) % 7 for i in range(4)]
    # 2. gradient contract: sum over ranks, caller applies 1/W
    flat = torch.full((1000,), float(rank + 1))
    scale = dp.all_reduce_sum_(flat)
    assert scale == 0.5 and torch.all(flat == 3.0)
    # 3. DP gradient == single-process gradient of the doubled batch
    cfg, params, data, tagsinfo = _tiny_problem()
    rows = np.arange(4)[rank::world]
    g_local, loss_local = _grads(cfg, params, data, tagsinfo, rows)
    sc = dp.all_reduce_sum_(g_local)
    g_dp = g_local * sc
    tot, cnt = dp.all_reduce_scalars([loss_local, 1.0])
    if rank == 0:
        g_full, loss_full = _grads(cfg, params, data, tagsinfo, np.arange(4))
        out.put((float((g_dp - g_full).abs().max()), float(g_full.abs().max()), tot / cnt, loss_full,
                 dp.steps_per_epoch(10, 4), dp.broadcast_object({"stop": False})))
    else:
        dp.broadcast_object(None)
    # 4. overlapped exchange protocol (kbner.dp.GradReducer): async buckets + host-agreed union of touched embedding rows
    #    == one dense all-reduce of the whole arena, for a sparse step, a dense step and the bf16-compressed variant
    import torch.distributed as dist

    class _TorchRows:   # CPU stand-ins for the HIP row kernels (gather / scatter / bf16 pack) -- test only
        gather_rows = staticmethod(lambda src, idx: src.index_select(0, idx.long()))
        scatter_rows = staticmethod(lambda rows, idx, dst: dst.index_copy_(0, idx.long(), rows))
        to_bf16 = staticmethod(lambda x: x.to(torch.bfloat16))
        from_bf16 = staticmethod(lambda y, out: out.copy_(y.float()))

    V, H, n = 64, 16, 5000
    lo = 2000
    checks = []
    for case, n_touch, compress in (("sparse", 5, False), ("dense", 60, False), ("dense_bf16", 60, True)):
        gen = torch.Generator().manual_seed(100 + rank)
        g = torch.randn(n, generator=gen)
        touched = torch.randperm(V, generator=gen)[:n_touch]
        emb = torch.zeros(V, H)
        emb[touched] = torch.randn(n_touch, H, generator=gen)
        g[lo:lo + V * H] = emb.flatten()
        want = g.clone()
        dist.all_reduce(want)
        flags = torch.zeros(V, dtype=torch.uint8)      # the optimizer's "row has received a gradient" flags
        flags[touched] = 3                             # what this rank's own embedding backward marks (LIVE | TOUCHED, include/kbner.h)
        red = dp.GradReducer(g, emb_range=(lo, lo + V * H), emb_width=H, compress_embedding=compress, row_ops=_TorchRows,
                             emb_flags=flags)
        red.begin(touched.numpy())
        red.bucket_ready(1000, 2000)     # the order backward finishes them: top layers first
        red.bucket_ready(0, 1000)
        scale = red.finish()
        tol = 2e-2 if compress else 1e-6
        # after the exchange every row that carries a gradient on ANY rank must be live here too (row-sparse AdamW relies on it)
        nz = (g[lo:lo + V * H].view(V, H) != 0).any(1)
        flags_ok = bool(torch.all(flags.bool() | ~nz)) and (case != "sparse" or int((flags != 0).sum()) <= 2 * n_touch)
        flags_ok = flags_ok and bool(torch.all((flags == 3) | (flags == 0)))   # a row the exchange wrote is live AND touched
        checks.append((case, red.stats["emb_mode"], float((g - want).abs().max()) <= tol * float(want.abs().max()) and flags_ok, scale,
                       red.stats["buckets"], red.stats["bytes_overlapped"]))
    # 4b. the A/B knobs of the first multi-GPU run (round 6: bench.py --bucket-layers / --exchange-delay): ready ranges coalesced two
    #     (and three: a non-contiguous range starts its own bucket) at a time, issue delayed by one bucket, the diagnostic blocking
    #     mode -- and a step WITHOUT any backward pass, where the owner's finalize hook must run before anything travels
    knob_checks = []
    for coalesce, delay, timed in ((2, 0, False), (1, 1, False), (2, 1, False), (3, 0, False)):
        gen = torch.Generator().manual_seed(300 + rank)
        g = torch.randn(n, generator=gen)
        want = g.clone()
        dist.all_reduce(want)
        red = dp.GradReducer(g, emb_range=None, coalesce=coalesce, delay=delay, time_buckets=timed)
        red.begin(None)
        for lo_, hi_ in ((4000, 5000), (3000, 4000), (1500, 2000), (1000, 1500), (0, 1000)):    # a gap between 3000 and 2000
            red.bucket_ready(lo_, hi_)
        red.finish()
        knob_checks.append((coalesce, delay, float((g - want).abs().max()) <= 1e-6 * float(want.abs().max()), red.stats["buckets"]))
    called = []
    g = torch.full((n,), float(rank + 1))
    red = dp.GradReducer(g, emb_range=None, finalize=lambda: (called.append(1), g[:100].zero_()))
    red.begin(None)
    red.finish()                     # no bucket was announced: g[:100] "still holds the previous step's gradients"
    knob_checks.append(("finalize", len(called), bool(torch.all(g[:100] == 0.0)) and bool(torch.all(g[100:] == 3.0)), 0))
    # 5. replicas start identical: broadcast of the parameter arena from rank 0
    p = torch.full((100,), float(rank + 7))
    dp.broadcast_params_(p)
    same = bool(torch.all(p == 7.0))
    if rank == 0:
        out.put(("reducer", checks, same, knob_checks))
    dp.barrier()
    dist.destroy_process_group()


def test_dp_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = out.get(timeout=240)
    res2 = out.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    tag, checks, same, knob_checks = res2
    assert tag == "reducer" and same
    # (coalesce, delay) -> buckets issued for five ready ranges with one gap: merged pairs / triples never span the gap
    assert [(c[0], c[1], c[3]) for c in knob_checks[:4]] == [(2, 0, 3), (1, 1, 5), (2, 1, 3), (3, 0, 3)], knob_checks
    assert all(c[2] for c in knob_checks), knob_checks
    assert knob_checks[4][:2] == ("finalize", 1), knob_checks
    modes = {c[0]: c[1] for c in checks}
    assert modes == {"sparse": "sparse", "dense": "dense", "dense_bf16": "dense_bf16"}, modes
    for case, mode, ok, scale, nb, nbytes in checks:
        assert ok and scale == 0.5 and nb == 2 and nbytes == 8000, (case, mode, ok, scale, nb, nbytes)
    err, gmax, loss_dp, loss_full, steps, obj = res
    assert err <= 2e-5 * max(gmax, 1.0), res          # same gradient up to fp32 summation order
    assert abs(loss_dp - loss_full) <= 1e-5 * abs(loss_full)
    assert steps == 2                                  # ceil(ceil(10/2)/4) global optimizer steps per epoch
    assert obj == {"stop": False}


def _worker_dense_fallback(rank, world, port, out):
    """world_size 4: every rank alone touches 20 % of the vocabulary (below sparse_threshold = 0.5), the UNION of the four
    exceeds it -> GradReducer must fall back to the dense embedding all-reduce and mark every row live (bench.py's synthetic
    workload at W = 8: ids uniform over the table, union ~88 % of it -- VERDICT round 3, weak #8)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
    torch.set_num_threads(1)
    import torch.distributed as dist
    from kbner import dp
    dp.init_from_env(backend="gloo")

    class _TorchRows:
        gather_rows = staticmethod(lambda src, idx: src.index_select(0, idx.long()))
        scatter_rows = staticmethod(lambda rows, idx, dst: dst.index_copy_(0, idx.long(), rows))
        to_bf16 = staticmethod(lambda x: x.to(torch.bfloat16))
        from_bf16 = staticmethod(lambda y, out: out.copy_(y.float()))

    V, H, lo = 200, 8, 300
    n = lo + V * H + 100
    res = []
    for case, per_rank in (("union_over_threshold", 40), ("union_under_threshold", 10)):
        gen = torch.Generator().manual_seed(7 + rank)
        g = torch.randn(n, generator=gen)
        touched = torch.arange(rank * per_rank, (rank + 1) * per_rank)     # disjoint per rank: union = world * per_rank rows
        emb = torch.zeros(V, H)
        emb[touched] = torch.randn(per_rank, H, generator=gen)
        g[lo:lo + V * H] = emb.flatten()
        want = g.clone()
        dist.all_reduce(want)
        flags = torch.zeros(V, dtype=torch.uint8)
        flags[touched] = 3
        red = dp.GradReducer(g, emb_range=(lo, lo + V * H), emb_width=H, row_ops=_TorchRows, emb_flags=flags)
        red.begin(touched.numpy())
        red.bucket_ready(0, lo)
        scale = red.finish()
        res.append((case, red.stats["emb_mode"], red.stats["emb_rows"], int((flags == 3).sum()), float((g - want).abs().max()), scale,
                    red.stats["bytes_tail"]))
    if rank == 0:
        out.put(res)
    dp.barrier()
    dist.destroy_process_group()


def test_dp_four_ranks_dense_embedding_fallback():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_dense_fallback, args=(r, 4, port, out)) for r in range(4)]
    for p in procs:
        p.start()
    res = out.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    V, H = 200, 8
    (c0, mode0, rows0, live0, err0, sc0, tail0), (c1, mode1, rows1, live1, err1, sc1, tail1) = res
    # 4 x 40 = 160 rows of 200 > 0.5 V: dense, every row marked live, the whole table (and the 100-float remainder) in the tail
    assert (mode0, rows0, live0) == ("dense", None, V) and err0 <= 1e-6 and sc0 == 0.25 and tail0 == 4 * (V * H + 100)
    # 4 x 10 = 40 rows < 0.5 V: sparse, exactly the union travels and is marked
    assert (mode1, rows1, live1) == ("sparse", 40, 40) and err1 <= 1e-6 and sc1 == 0.25 and tail1 == 4 * (40 * H + 100)


def test_shard_indices_properties():
    sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))
    from kbner import dp
    for n in (1, 5, 8, 13):
        for w in (1, 2, 4, 8):
            parts = [dp.shard_indices(n, r, w) for r in range(w)]
            assert len({len(p) for p in parts}) == 1                   # equal step counts
            assert set(i for p in parts for i in p) == set(range(n))   # every batch is visited
