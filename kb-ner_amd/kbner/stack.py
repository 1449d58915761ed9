"""Inference engine of BASELINE config 5 (ACE-style stacked embeddings -> BiLSTM -> linear -> CRF Viterbi), all arithmetic in
libkbner_hip.so.

Reference path (restated): FastSequenceTagger.forward with `use_rnn` (flair/models/sequence_tagger_model.py:844-1052):
    sentence_tensor = cat([features[name] * selection[idx] for idx, name in enumerate(sorted(features))], -1)   (:879-891)
    packed BiLSTM (torch.nn.LSTM(D, hidden, 1 layer, bidirectional), :324-357, :969-994) -> linear(2 * hidden -> T) (:1027)
and the feature producers: TransformerWordEmbeddings (frozen, first-sub-token pooling, optionally `use_internal_doc`:
embeddings.py:3116-3117,3283-3284) and FlairEmbeddings (character LM hidden state at each token's end,
embeddings.py:2469-2543, flair/models/language_model.py:71-138).

Layout: ONE bf16 matrix X [rows = B * n padded to 128, D_total padded to 64] is the concatenation; every producer writes its own
column block in place (transformers: kbner_gather_rows_ld from the encoder's hidden states; character LMs: the LSTM step
kernel stores h_t at the token-end steps straight into X), so torch.cat never happens.  The BiLSTM's input half is one MFMA
GEMM over all time steps, its recurrent half one launch per time step for both directions (csrc/lstm.hip)."""
import numpy as np
import torch

from . import lib as L_
from . import ops
from .lib import EPI_BIAS, GEMM_NT

BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32


def _round_up(x, m):
    return (x + m - 1) // m * m


class LSTMGroup:
    """`ndir` single-layer LSTMs of one hidden width run in lockstep (the two directions of the tagger's BiLSTM; one character
    LM).  Holds Whh as bf16 [ndir, 4Hp, Hp] (hidden zero-padded to a multiple of 64: a padded unit has zero weights, so its c
    and h stay exactly 0) and runs the recurrence given pre-activations and per-step row tables."""

    def __init__(self, w_hh_list, hidden, device):
        self.H = int(hidden)
        self.Hp = _round_up(self.H, 64)   # kbner_lstm_seq: a wave's k slice is 64 wide (whole 128-byte lines of Whh)
        self.ndir = len(w_hh_list)
        self.device = torch.device(device)
        whh = torch.zeros((self.ndir, 4 * self.Hp, self.Hp), dtype=F32)
        for d, w in enumerate(w_hh_list):
            w = torch.as_tensor(w, dtype=F32)
            assert tuple(w.shape) == (4 * self.H, self.H)
            for q in range(4):
                whh[d, q * self.Hp:q * self.Hp + self.H, :self.H] = w[q * self.H:(q + 1) * self.H]
        self.whh = whh.to(BF16).to(self.device).contiguous()

    def pad_gates(self, t, axis=0):
        """[4H, ...] gate-major (i|f|g|o) tensor -> [4Hp, ...] with every gate block zero-padded"""
        t = torch.as_tensor(t, dtype=F32)
        out = torch.zeros((4 * self.Hp,) + tuple(t.shape[1:]), dtype=F32)
        for q in range(4):
            out[q * self.Hp:q * self.Hp + self.H] = t[q * self.H:(q + 1) * self.H]
        return out

    def run(self, gx, gxi, outi, out, out_dir_stride, B, col=0, out_cols=None):
        """gx bf16 [rows, ndir*4Hp]; gxi / outi int32 [steps, ndir, B] (device); out bf16 [rows_out, ld]; col: first column of
        direction 0's h inside `out`, direction d at col + d * out_dir_stride -- or out_cols: one first column per direction.
        The whole recurrence is ONE call into the library (kbner_lstm_seq: one launch per time step, enqueued back to back)."""
        steps = gxi.shape[0]
        if out_cols is None:
            out_cols = [col + d * out_dir_stride for d in range(self.ndir)]
        if len(out_cols) != self.ndir or any(int(x) % 4 for x in out_cols):
            raise L_.KbnerError("LSTM output column offsets: one per direction, each a multiple of 4")
        h = torch.zeros((2, self.ndir, B, self.Hp), dtype=BF16, device=self.device)
        c = torch.zeros((self.ndir, B, self.Hp), dtype=F32, device=self.device)
        oc = torch.tensor([int(x) for x in out_cols], dtype=I32, device=self.device)
        gxi, outi = gxi.contiguous(), outi.contiguous()
        L_.call("kbner_lstm_seq", L_.ptr(gx), gx.shape[-1], L_.ptr(gxi), L_.ptr(self.whh), L_.ptr(h), L_.ptr(c), L_.ptr(out),
                out.shape[-1], L_.ptr(oc), L_.ptr(outi), steps, B, self.Hp, self.ndir, L_.stream_ptr())
        return h[steps & 1]

    def run_stepwise(self, gx, gxi, outi, out, out_dir_stride, B, col=0):
        """the same recurrence through the one-wave-per-tile step kernel, one library call per time step (kept as the A/B and
        cross-check of kbner_lstm_seq)"""
        steps = gxi.shape[0]
        h = [torch.zeros((self.ndir, B, self.Hp), dtype=BF16, device=self.device) for _ in range(2)]
        c = torch.zeros((self.ndir, B, self.Hp), dtype=F32, device=self.device)
        if col % 4:
            raise L_.KbnerError("LSTM output column offset must be a multiple of 4")
        for s in range(steps):
            self._step(gx, gxi[s], h[s & 1], h[(s + 1) & 1], c, out, col, outi[s], out_dir_stride)
        return h[steps & 1]

    def _step(self, gx, gxi, h_in, h_out, c, out, col, outi, out_dir_stride):
        ndir, B, Hp = h_in.shape
        L_.call("kbner_lstm_step", L_.ptr(gx), gx.shape[-1], L_.ptr(gxi), L_.ptr(self.whh), L_.ptr(h_in), L_.ptr(h_out), L_.ptr(c),
                L_.c_void_p(out.data_ptr() + 2 * col), out.shape[-1], out_dir_stride, L_.ptr(outi), B, Hp, ndir, L_.stream_ptr())


def step_tables(lengths, n, bidirectional=True):
    """row tables of a packed (variable-length) pass over token-major rows b*n + t: forward direction consumes t = s, the
    backward direction t = len_b - 1 - s; finished sequences get -1 (torch pack_padded_sequence / pad_packed_sequence: state
    frozen, padded outputs stay zero).  -> int32 [steps, ndir, B]"""
    lengths = np.asarray(lengths, np.int64)
    B = len(lengths)
    steps = int(lengths.max()) if B else 0
    s = np.arange(steps)[:, None]
    base = (np.arange(B) * n)[None, :]
    live = s < lengths[None, :]
    fwd = np.where(live, base + s, -1)
    if not bidirectional:
        return fwd[:, None, :].astype(np.int32)
    bwd = np.where(live, base + (lengths[None, :] - 1 - s), -1)
    return np.stack([fwd, bwd], 1).astype(np.int32)


class BiLSTMHead:
    """BiLSTM(D -> hidden, 1 layer) + linear(2*hidden -> T) on the concatenated features (sequence_tagger_model.py:969-1027)."""

    def __init__(self, rnn_state, linear_w, linear_b, blocks, hidden, device):
        """rnn_state: torch.nn.LSTM state dict (weight_ih_l0, weight_hh_l0, bias_ih_l0, bias_hh_l0 and *_reverse);
        blocks: the widths D_i of the concatenated feature blocks in the reference's order (sorted embedding names).  Inside X
        every block starts at a multiple of 32 columns (`self.cols[i]`), so a producer whose width is padded (a 1000-unit LM
        writes 1024 columns, the last 24 zero) never touches its neighbour; Wih's columns are placed to match."""
        self.device = torch.device(device)
        self.blocks = [int(w) for w in blocks]
        self.D, self.H = sum(self.blocks), int(hidden)
        self.cols, off = [], 0
        for w in self.blocks:
            self.cols.append(off)
            off += _round_up(w, 32)
        self.Dp = _round_up(max(off, 1), 64)
        self.grp = LSTMGroup([rnn_state["weight_hh_l0"], rnn_state["weight_hh_l0_reverse"]], hidden, device)
        Hp = self.Hp = self.grp.Hp
        wih = torch.zeros((8 * Hp, self.Dp), dtype=F32)
        bias = torch.zeros(8 * Hp, dtype=F32)
        for d, sfx in enumerate(("", "_reverse")):
            w = self.grp.pad_gates(rnn_state["weight_ih_l0" + sfx])
            assert w.shape[1] == self.D, (w.shape, self.D)
            ref = 0
            for width, xc in zip(self.blocks, self.cols):
                wih[d * 4 * Hp:(d + 1) * 4 * Hp, xc:xc + width] = w[:, ref:ref + width]
                ref += width
            bias[d * 4 * Hp:(d + 1) * 4 * Hp] = self.grp.pad_gates(torch.as_tensor(rnn_state["bias_ih_l0" + sfx], dtype=F32)
                                                                   + torch.as_tensor(rnn_state["bias_hh_l0" + sfx], dtype=F32))
        self.wih = wih.to(BF16).to(self.device).contiguous()
        self.bias = bias.to(self.device)
        lw = torch.as_tensor(linear_w, dtype=F32)
        T = lw.shape[0]
        wl = torch.zeros((T, 2 * Hp), dtype=F32)
        wl[:, :self.H] = lw[:, :self.H]
        wl[:, Hp:Hp + self.H] = lw[:, self.H:2 * self.H]
        self.lin_w = wl.to(self.device).contiguous()
        self.lin_b = torch.as_tensor(linear_b, dtype=F32).to(self.device).contiguous()
        self.T = T

    def rows(self, B, n):
        return _round_up(max(B * n, 1), 128)

    def new_input(self, B, n):
        """zeroed X bf16 [rows, Dp] the feature producers fill"""
        return torch.zeros((self.rows(B, n), self.Dp), dtype=BF16, device=self.device)

    def emissions(self, X, lengths, B, n):
        """X bf16 [rows, Dp] (token-major rows b*n + t, zero rows at padding) -> emissions f32 [B, n, T]"""
        with L_.stream_scope():
            Mp, Hp = X.shape[0], self.Hp
            gx = torch.empty((Mp, 8 * Hp), dtype=BF16, device=self.device)
            ops.gemm(GEMM_NT, X, self.wih, Mp, 8 * Hp, self.Dp, C=gx, bias=self.bias, epi=EPI_BIAS)
            tab = torch.from_numpy(step_tables(lengths, n)).to(self.device)
            out = torch.zeros((Mp, 2 * Hp), dtype=BF16, device=self.device)
            self.grp.run(gx, tab, tab, out, Hp, B)
            em = ops.head_fwd(out[:B * n], self.lin_w, self.lin_b)
        return em.view(B, n, self.T)


class CharLM:
    """FlairEmbeddings' character language model at inference: embedding(chars) -> 1-layer LSTM -> hidden state at chosen steps
    (flair/models/language_model.py:71-138; projection `nout` unsupported: the shipped big LMs have none).
    The input half is a lookup table: table[ch] = Wih emb[ch] + bih + bhh, one row per dictionary character, built once."""

    def __init__(self, state_dict, hidden, device):
        self.device = torch.device(device)
        self.grp = LSTMGroup([state_dict["rnn.weight_hh_l0"]], hidden, device)
        emb = torch.as_tensor(state_dict["encoder.weight"], dtype=F32)
        wih = torch.as_tensor(state_dict["rnn.weight_ih_l0"], dtype=F32)
        b = torch.as_tensor(state_dict["rnn.bias_ih_l0"], dtype=F32) + torch.as_tensor(state_dict["rnn.bias_hh_l0"], dtype=F32)
        table = emb @ wih.t() + b                                   # [chars, 4H] fp32, host, once per model load
        self.table = self.grp.pad_gates(table.t().contiguous()).t().contiguous().to(BF16).to(self.device).contiguous()
        self.H, self.Hp = self.grp.H, self.grp.Hp

    def run(self, char_ids, out_rows, X, col):
        """char_ids int [steps, B] (every sequence padded to the same length, as the reference pads with blanks);
        out_rows int [steps, B]: row of X that receives h after that step, -1 = not needed.  Writes X[row, col:col+Hp]."""
        steps, B = char_ids.shape
        gxi = torch.from_numpy(np.ascontiguousarray(char_ids, np.int32)[:, None, :]).to(self.device)
        outi = torch.from_numpy(np.ascontiguousarray(out_rows, np.int32)[:, None, :]).to(self.device)
        with L_.stream_scope():
            self.grp.run(self.table, gxi, outi, X, 0, B, col=col)


class CharLMGroup:
    """All character LMs of one hidden width run as ONE recurrence (LSTMGroup with ndir = number of models): one launch per
    character step for all of them instead of one per model -- the step of a 2048-unit LM is a 33.5 MB stream of Whh, and a
    single model's launch cannot keep enough of it in flight.  Tables are stacked along the columns ([chars, ndir * 4Hp], the
    shorter dictionaries zero-padded), every model keeps its own character ids (forward / backward LMs read the text in opposite
    orders) and its own column block of X."""

    def __init__(self, lms):
        lms = list(lms)
        if not lms or len({lm.Hp for lm in lms}) != 1:
            raise ValueError("a CharLMGroup takes character LMs of one (padded) hidden width")
        self.lms = lms
        self.device = lms[0].device
        self.H, self.Hp = lms[0].H, lms[0].Hp
        self.grp = LSTMGroup.__new__(LSTMGroup)
        self.grp.H, self.grp.Hp, self.grp.ndir, self.grp.device = self.H, self.Hp, len(lms), self.device
        self.grp.whh = torch.cat([lm.grp.whh for lm in lms], 0).contiguous()
        rows = max(lm.table.shape[0] for lm in lms)
        table = torch.zeros((rows, len(lms) * 4 * self.Hp), dtype=BF16, device=self.device)
        for d, lm in enumerate(lms):
            table[:lm.table.shape[0], d * 4 * self.Hp:(d + 1) * 4 * self.Hp] = lm.table
        self.table = table

    def run(self, char_ids, out_rows, X, cols):
        """char_ids / out_rows: one int [steps, B] array per model (same steps and B); cols: its first column in X"""
        gxi = torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(c, np.int32) for c in char_ids], 1))).to(self.device)
        outi = torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(r, np.int32) for r in out_rows], 1))).to(self.device)
        B = gxi.shape[2]
        with L_.stream_scope():
            self.grp.run(self.table, gxi, outi, X, 0, B, out_cols=list(cols))
