"""TEST INFRASTRUCTURE.  CPU restatement (numpy, fp32) of BASELINE config 5's inference stack, pinned by tests/golden/stack.*
(captured by running the reference, oracle/gen_golden_stack.py):

  lstm_layer            torch.nn.LSTM semantics, one layer, one direction (gate order i|f|g|o; the reference's modules are
                        torch.nn.LSTM: flair/models/sequence_tagger_model.py:340-346, flair/models/language_model.py:41-44)
  bilstm_packed         pack_padded_sequence -> bidirectional LSTM -> pad_packed_sequence (:969-984): every sequence runs over
                        its own length, the backward direction starts at its last real token, padded outputs are zero
  flair_features        FlairEmbeddings._add_embeddings_internal (flair/embeddings.py:2469-2543): "\\n" + text (reversed for a
                        backward LM) + " " blank-padded, hidden state taken at each token's end offset
  stack_emissions       FastSequenceTagger.forward with use_rnn (:879-891 selection-masked concat in sorted-name order, :969-1027)
Imported only by tests/ (and bench / smoke checkers); never by the product path."""
import numpy as np


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def lstm_layer(x, w_ih, w_hh, b_ih, b_hh, h0=None, c0=None):
    """x f32 [steps, B, D] -> (out f32 [steps, B, H], (h, c))"""
    x = np.asarray(x, np.float32)
    steps, B, _ = x.shape
    H = w_hh.shape[1]
    h = np.zeros((B, H), np.float32) if h0 is None else h0.astype(np.float32)
    c = np.zeros((B, H), np.float32) if c0 is None else c0.astype(np.float32)
    out = np.zeros((steps, B, H), np.float32)
    for t in range(steps):
        g = x[t] @ w_ih.T + b_ih + h @ w_hh.T + b_hh
        i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
        c = _sigmoid(f) * c + _sigmoid(i) * np.tanh(gg)
        h = _sigmoid(o) * np.tanh(c)
        out[t] = h
    return out, (h, c)


def bilstm_packed(x, lengths, rnn):
    """x f32 [B, n, D], lengths int[B], rnn: torch.nn.LSTM state dict as numpy -> f32 [B, n, 2H] (zeros at padding)"""
    B, n, _ = x.shape
    H = rnn["weight_hh_l0"].shape[1]
    out = np.zeros((B, n, 2 * H), np.float32)
    for b in range(B):
        L = int(lengths[b])
        seq = x[b, :L][:, None, :]
        f, _ = lstm_layer(seq, rnn["weight_ih_l0"], rnn["weight_hh_l0"], rnn["bias_ih_l0"], rnn["bias_hh_l0"])
        r, _ = lstm_layer(seq[::-1], rnn["weight_ih_l0_reverse"], rnn["weight_hh_l0_reverse"], rnn["bias_ih_l0_reverse"],
                          rnn["bias_hh_l0_reverse"])
        out[b, :L, :H] = f[:, 0]
        out[b, :L, H:] = r[::-1, 0]
    return out


def flair_features(sentences, lm, chars, is_forward):
    """sentences: list of token-text lists; lm: LanguageModel state dict as numpy; chars: the LM dictionary's items in order
    (index 0 = <unk>).  -> f32 [B, n, H]"""
    idx = {c: i for i, c in enumerate(chars)}
    texts = [" ".join(s) for s in sentences]
    longest = max(len(t) for t in texts)
    B, n = len(sentences), max(len(s) for s in sentences)
    framed = ["\n" + (t if is_forward else t[::-1]) + " " + " " * (longest - len(t)) for t in texts]
    ids = np.asarray([[idx.get(ch, 0) for ch in f] for f in framed], np.int64).T      # [steps, B]
    emb = lm["encoder.weight"][ids]                                                  # [steps, B, E]
    hs, _ = lstm_layer(emb, lm["rnn.weight_ih_l0"], lm["rnn.weight_hh_l0"], lm["rnn.bias_ih_l0"], lm["rnn.bias_hh_l0"])
    H = hs.shape[-1]
    out = np.zeros((B, n, H), np.float32)
    for b, (s, t) in enumerate(zip(sentences, texts)):
        off_f, off_b = 1, len(t) + 1
        for k, tok in enumerate(s):
            off_f += len(tok)
            out[b, k] = hs[off_f if is_forward else off_b, b]
            off_f += 1
            off_b -= 1 + len(tok)
    return out


def stack_emissions(features_by_name, selection, lengths, rnn, lin_w, lin_b):
    """features_by_name: {embedding name: f32 [B, n, D_i]}; selection: 0/1 per sorted name -> emissions f32 [B, n, T]"""
    names = sorted(features_by_name)
    x = np.concatenate([features_by_name[nm] * np.float32(selection[i]) for i, nm in enumerate(names)], -1)
    h = bilstm_packed(x, lengths, rnn)
    return h @ lin_w.T + lin_b
