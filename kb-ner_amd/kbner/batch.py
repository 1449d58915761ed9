"""Host-side (numpy) batch assembly: integer metadata only -- position ids, additive mask, flat
row indices for first-subtoken pooling, and the remove_x compaction index.  Mirrors what
TransformerWordEmbeddings._add_embeddings_to_sentences (flair/embeddings.py:3247-3260),
ColumnDataLoader.assign_tags (flair/custom_data_loader.py:356-374) and
FastSequenceTagger._calculate_loss's remove_x branch (sequence_tagger_model.py:2448-2488) compute
per batch, moved off the device hot path."""
import numpy as np
import torch

SEED = 20220711


def round_up(x, m):
    return (x + m - 1) // m * m


def assemble(input_ids, attention_mask, first_idx, tags, lengths, x_idx, pad_id=1, s_multiple=64, first_row=None,
             position_mode="roberta"):
    """input_ids/attention_mask int[R,S0] (R encoder rows; R == B unless sentences were split into sliding windows);
    first_idx int[B,n] (sub-token position of each word token's first piece inside its row, -1 = none/padding);
    first_row int[B,n] (encoder row of that piece; default: row b); tags int[B,n]; lengths int[B] (word tokens per
    sentence).  Returns a dict of numpy arrays (see to_device)."""
    input_ids = np.asarray(input_ids, np.int64)
    attention_mask = np.asarray(attention_mask, np.int64)
    first_idx = np.asarray(first_idx, np.int64)
    tags = np.asarray(tags, np.int64)
    lengths = np.asarray(lengths, np.int64)
    R, S0 = input_ids.shape
    B = first_idx.shape[0]
    n = first_idx.shape[1]
    S = max(64, round_up(S0, s_multiple))
    if S > 512:
        raise ValueError("sequence length %d exceeds the encoder's 512-position window" % S0)
    ids = np.zeros((R, S), np.int64)            # the reference pads ids with 0 and the mask with 0 (:3247-3260)
    am = np.zeros((R, S), np.int64)
    ids[:, :S0] = input_ids
    am[:, :S0] = attention_mask
    if position_mode == "roberta":
        nz = (ids != pad_id).astype(np.int64)   # RoBERTa position ids from ids != pad (transformers modeling_roberta)
        pos = np.cumsum(nz, axis=1) * nz + pad_id
    elif position_mode == "absolute":           # BERT: position = index in the row (transformers modeling_bert)
        pos = np.tile(np.arange(S, dtype=np.int64)[None, :], (R, 1))
    else:
        raise ValueError("position_mode must be 'roberta' or 'absolute'")
    M = R * S
    Mp = round_up(M, 256)
    ids_f = np.zeros(Mp, np.int32)
    pos_f = np.full(Mp, pad_id if position_mode == "roberta" else 0, np.int32)
    ids_f[:M] = ids.reshape(-1)
    pos_f[:M] = pos.reshape(-1)
    maskbias = ((1 - am) * -10000.0).astype(np.float32)
    rows = np.arange(B, dtype=np.int64)[:, None] if first_row is None else np.asarray(first_row, np.int64)
    row_idx = np.where(first_idx >= 0, rows * S + first_idx, -1).astype(np.int32)
    valid = np.arange(n)[None, :] < lengths[:, None]
    keep = valid & (tags != x_idx) if x_idx is not None else valid
    clens = keep.sum(axis=1).astype(np.int32)
    nc = max(1, int(clens.max()) if B else 1)
    crow = np.full((B, nc), -1, np.int32)
    ctags = np.zeros((B, nc), np.int32)
    cpos = np.full((B, nc), -1, np.int32)       # word-token position of each compacted row (WordDropout drops positions)
    cfeat = np.full((B, nc), -1, np.int32)      # row of the [B * n, T] all-token emissions each compacted row comes from
    for b in range(B):
        k = np.nonzero(keep[b])[0]
        crow[b, :len(k)] = row_idx[b, k]
        ctags[b, :len(k)] = tags[b, k]
        cpos[b, :len(k)] = k
        cfeat[b, :len(k)] = b * n + k
    return dict(cfeat_idx=cfeat.reshape(-1), keep_f=keep.astype(np.float32), B=B, R=R, S=S, ids=ids_f, pos_ids=pos_f, maskbias=maskbias, row_idx=row_idx.reshape(-1), lengths=lengths.astype(np.int32),
                tags=tags.astype(np.int32), keep=keep, crow_idx=crow.reshape(-1), ctags=ctags, clens=clens,
                cpos=cpos.reshape(-1), n_tokens=n,
                input_ids=ids, attention_mask=am, first_idx=first_idx)


_DEVICE_KEYS = ("ids", "pos_ids", "maskbias", "row_idx", "lengths", "tags", "crow_idx", "ctags", "clens", "cpos", "cfeat_idx", "keep_f")


def to_device(batch, device="cuda"):
    out = {"B": batch["B"], "R": batch.get("R", batch["B"]), "S": batch["S"], "n_tokens": batch["n_tokens"]}
    for k in _DEVICE_KEYS:
        out[k] = torch.from_numpy(np.ascontiguousarray(batch[k])).to(device)
    return out


def synthetic_sentences(B, S=512, vocab=250002, T=29, x_idx=9, start=27, stop=28, n_real=16, seed=SEED):
    """SURVEY.md §8(d) synthetic workload: <s> + 510 content ids uniform in [5, vocab) + </s>; word
    structure: n_real real tokens, one <EOS> token, then context tokens, sub-tokens/token drawn from
    {1: .7, 2: .2, 3: .1} until the 510 content slots are filled; gold tags of real tokens uniform over
    the non-special non-X tags, <EOS> + context tags = S-X."""
    rng = np.random.default_rng(seed)
    content = S - 2
    ids = rng.integers(5, vocab, size=(B, S), dtype=np.int64)
    ids[:, 0] = 0
    ids[:, -1] = 2
    am = np.ones((B, S), np.int64)
    valid_tags = np.asarray([t for t in range(T) if t not in (0, x_idx, start, stop)], np.int64)
    firsts, tag_rows = [], []
    for b in range(B):
        pos, f = 1, []
        while pos < 1 + content:
            f.append(pos)
            pos += int(rng.choice([1, 2, 3], p=[0.7, 0.2, 0.1]))
        firsts.append(f)
        tg = np.full(len(f), x_idx, np.int64)
        k = min(n_real, len(f))
        tg[:k] = rng.choice(valid_tags, size=k)
        tag_rows.append(tg)
    n = max(len(f) for f in firsts)
    first_idx = np.full((B, n), -1, np.int64)
    tags = np.zeros((B, n), np.int64)
    lengths = np.zeros(B, np.int64)
    for b in range(B):
        L = len(firsts[b])
        first_idx[b, :L] = firsts[b]
        tags[b, :L] = tag_rows[b]
        lengths[b] = L
    return ids, am, first_idx, tags, lengths


def synthetic_batch(B, S=512, vocab=250002, T=29, x_idx=9, start=27, stop=28, n_real=16, seed=SEED):
    ids, am, first_idx, tags, lengths = synthetic_sentences(B, S, vocab, T, x_idx, start, stop, n_real, seed)
    return assemble(ids, am, first_idx, tags, lengths, x_idx)
