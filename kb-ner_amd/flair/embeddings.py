"""Embeddings surface of the tagger path: Embeddings / TokenEmbeddings / StackedEmbeddings / TransformerWordEmbeddings.

Behavioural reference (restated): flair/embeddings.py -- Embeddings.embed (:75-101), StackedEmbeddings (:155-211),
TransformerWordEmbeddings (:2906-3907): tokenizer + encoder loaded from a directory, `<EOS>` word tokens replaced by the
tokenizer's eos string before sub-tokenisation (:3139-3163), sub-token counts per word token recovered by re-assembling
sub-token text (:3347-3408), counts clamped to `maximum_subtoken_length` (:3182-3195), `<s> ids </s>` framing with
`begin_offset = 1` (:3202-3227), ids AND mask padded with 0 (:3247-3260), first-sub-token pooling of the LAST layer
(:3288-3345) with zero vectors for word tokens that received no sub-token (:3306-3308).

Difference by design: nothing numeric happens here.  The reference runs the HF encoder inside this class and bounces
[B,n,H] features through the CPU; here the class only produces the INTEGER batch (ids, mask, first-sub-token index) and
the encoder itself lives in the tagger's HIP engine (kbner.engine.Tagger), which keeps hidden states on the device.
Only the configuration the KB-NER YAMLs use is implemented: layers '-1', pooling_operation 'first', single 512 window
(inputs are cut to <= 510 sub-tokens at data-generation time, kb/context_process.py:974); anything else raises."""
import json
import logging
import os
import re
from typing import List

import numpy as np
import torch

import flair

log = logging.getLogger("flair")


class Embeddings(torch.nn.Module):
    @property
    def embedding_length(self) -> int:
        raise NotImplementedError

    @property
    def embedding_type(self) -> str:
        return "word-level"

    def embed(self, sentences, embedding_mask=None):
        if not isinstance(sentences, list) and not hasattr(sentences, "features"):
            sentences = [sentences]
        self._add_embeddings_internal(sentences)
        return sentences

    def _add_embeddings_internal(self, sentences):
        raise NotImplementedError


class TokenEmbeddings(Embeddings):
    pass


class StackedEmbeddings(TokenEmbeddings):
    def __init__(self, embeddings: List[TokenEmbeddings], gpu_friendly=False):
        super().__init__()
        self.embeddings = embeddings
        for i, e in enumerate(embeddings):
            self.add_module("list_embedding_%d" % i, e)
        self.name = "Stack"
        self.static_embeddings = all(getattr(e, "static_embeddings", False) for e in embeddings)
        self.gpu_friendly = gpu_friendly

    @property
    def embedding_length(self) -> int:
        return sum(e.embedding_length for e in self.embeddings)

    def embed(self, sentences, static_embeddings: bool = True, embedding_mask=None):
        if not isinstance(sentences, list) and not hasattr(sentences, "features"):
            sentences = [sentences]
        for e in self.embeddings:
            e.embed(sentences)
        return sentences

    def _add_embeddings_internal(self, sentences):
        for e in self.embeddings:
            e._add_embeddings_internal(sentences)
        return sentences

    def __str__(self):
        return "StackedEmbeddings [%s]" % ",".join(str(e) for e in self.embeddings)


class _EncoderHandle:
    """What train.py / the trainer touch on `embedding.model`: `.config`, `.save_pretrained(dir)`, `.to()`, `.eval()`.
    Weights are held as a HF-named state dict until the tagger moves them into its arena; afterwards `source` points at
    the live engine so save_pretrained writes the CURRENT fine-tuned weights."""

    def __init__(self, config, state_dict):
        self.config = config
        self._state_dict = state_dict
        self.source = None  # kbner.engine.Tagger once attached

    def state_dict(self):
        if self.source is not None:
            return {k: v.detach().float().cpu() for k, v in self.source.hf_state_dict().items()}
        return self._state_dict

    def save_pretrained(self, save_directory):
        os.makedirs(save_directory, exist_ok=True)
        cfg = self.config.to_dict() if hasattr(self.config, "to_dict") else dict(self.config)
        with open(os.path.join(save_directory, "config.json"), "w") as f:
            json.dump(cfg, f, indent=2, sort_keys=True, default=str)
        sd = {"roberta." + k if False else k: v.contiguous() for k, v in self.state_dict().items()}
        try:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(save_directory, "model.safetensors"), metadata={"format": "pt"})
        except Exception:
            torch.save(sd, os.path.join(save_directory, "pytorch_model.bin"))

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def train(self, mode=True):
        return self


def _load_hf_dir(model_dir):
    """(config, HF-named fp32 state dict) from a local HF directory: config.json + model.safetensors | pytorch_model.bin."""
    from transformers import AutoConfig
    config = AutoConfig.from_pretrained(model_dir)
    st = os.path.join(model_dir, "model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        sd = load_file(st)
    else:
        sd = torch.load(os.path.join(model_dir, "pytorch_model.bin"), map_location="cpu", weights_only=True)
    out = {}
    for k, v in sd.items():
        for pre in ("roberta.", "xlm_roberta.", "bert."):
            if k.startswith(pre):
                k = k[len(pre):]
        out[k] = v.float()
    return config, out


class TransformerWordEmbeddings(TokenEmbeddings):
    def __init__(self, model: str = "xlm-roberta-large", layers: str = "-1", pooling_operation: str = "first", batch_size: int = 1,
                 use_scalar_mix: bool = False, fine_tune: bool = False, allow_long_sentences: bool = True, stride: int = -1,
                 maximum_window: bool = False, document_extraction: bool = False, embedding_name: str = None,
                 doc_batch_size: int = 32, maximum_subtoken_length: int = 999, v2_doc: bool = False, ext_doc: bool = False,
                 sentence_feat: bool = False, use_internal_doc: bool = False, **kwargs):
        super().__init__()
        os.environ["TOKENIZERS_PARALLELISM"] = "false"
        if not os.path.isdir(model):
            raise FileNotFoundError("TransformerWordEmbeddings(model=%r): a local HF directory is required (no network); it must "
                                    "hold the tokenizer files, config.json and model.safetensors / pytorch_model.bin" % model)
        if [int(x) for x in str(layers).split(",")] != [-1] or pooling_operation != "first" or use_scalar_mix:
            raise NotImplementedError("only layers='-1', pooling_operation='first' (the KB-NER configs) are on the MI355X path")
        if document_extraction or ext_doc or sentence_feat:
            raise NotImplementedError("document_extraction / ext_doc / sentence_feat are outside the hot path (SURVEY.md §8f-3); "
                                      "v2_doc (document windows) is supported")
        self.v2_doc = bool(v2_doc)        # also settable later: train.py --v2doc does `embedding.v2_doc = True` (:223-224)
        self.doc_batch_size = doc_batch_size
        from transformers import AutoTokenizer
        self.tokenizer = AutoTokenizer.from_pretrained(model, **kwargs)
        config, sd = _load_hf_dir(model)
        self.model = _EncoderHandle(config, sd)
        self.name = str(model) if embedding_name is None else embedding_name
        self.layer_indexes = [-1]
        self.pooling_operation = pooling_operation
        self.use_scalar_mix = False
        self.fine_tune = fine_tune
        self.static_embeddings = not fine_tune
        self.batch_size = batch_size
        self.sentence_feat = False
        self.use_internal_doc = use_internal_doc
        self.allow_long_sentences = allow_long_sentences
        self.max_subtokens_sequence_length = min(512, int(getattr(self.tokenizer, "model_max_length", 512) or 512))
        self.stride = self.max_subtokens_sequence_length // 2 if allow_long_sentences else 0
        self.begin_offset = 1
        self.maximum_subtoken_length = maximum_subtoken_length
        bos = getattr(self.tokenizer, "bos_token", None) or getattr(self.tokenizer, "cls_token", None)
        eos = getattr(self.tokenizer, "eos_token", None) or getattr(self.tokenizer, "sep_token", None)
        if bos is None or eos is None:
            raise ValueError("tokenizer needs bos/cls and eos/sep tokens")
        self._bos_id = self.tokenizer.convert_tokens_to_ids(bos)
        self._eos_id = self.tokenizer.convert_tokens_to_ids(eos)
        self._eos_text = eos
        self.special_tokens = [t for t in (bos, getattr(self.tokenizer, "cls_token", None)) if t]
        self._hidden = int(config.hidden_size)

    @property
    def embedding_length(self) -> int:
        return self._hidden

    # ------------------------------------------------------------------ tokenisation (host, integers only)
    @staticmethod
    def _strip_markup(piece: str) -> str:
        """drop the word-boundary markers sub-word tokenizers prepend / append (sentencepiece, BERT, byte-level BPE, XLM)"""
        for pre in ("Ġ", "##", "▁"):
            if piece.startswith(pre):
                piece = piece[len(pre):]
        return piece[:-4] if piece.endswith("</w>") else piece

    def _token_text(self, token) -> str:
        """the word token as the tokenizer spells it (its own pieces re-joined, lower-cased): unknown / normalised
        characters then compare equal between the per-token and the per-sentence tokenisation (embeddings.py:3103-3109)"""
        text = token.text
        cache = self.__dict__.setdefault("_token_text_cache", {})
        hit = cache.get(text)
        if hit is None:
            hit = "".join(self._strip_markup(p) for p in self.tokenizer.tokenize(text)).lower()
            if len(cache) < 1_000_000:  # word types, not tokens: bounded by the corpus vocabulary
                cache[text] = hit
        return hit

    def reconstruct_tokens_from_subtokens(self, tokens, subtokens) -> List[int]:
        """Sub-token count per word token, by re-assembling sub-token text against the word tokens in order.
        A word token the tokenizer dropped entirely gets 0 (it will pool to a zero vector)."""
        texts = [self._token_text(t) for t in tokens]
        counts: List[int] = []
        ti, acc, cnt = 0, "", 0
        for si, piece in enumerate(subtokens):
            if ti >= len(texts):
                break
            piece = self._strip_markup(piece)
            if si == 0 and piece in self.special_tokens:
                continue
            low = piece.lower()
            if cnt == 0 and not texts[ti].startswith(low):
                # the tokenizer skipped one or more word tokens: give them 0 pieces and resynchronise
                while ti < len(texts) and not texts[ti].startswith(low):
                    counts.append(0)
                    ti += 1
                if ti >= len(texts):
                    break
            cnt += 1
            acc += low
            if acc == texts[ti]:
                counts.append(cnt)
                ti, acc, cnt = ti + 1, "", 0
        while len(counts) < len(texts):
            counts.append(cnt if cnt and len(counts) == ti else 0)
            cnt = 0
        return counts

    def tokenize_sentence(self, sentence):
        """cached per Sentence object: corpora are static across epochs, and at ~1 ms of tokenizer time per 512-piece
        sentence re-tokenising every batch would make the host the bottleneck of a 40-ms GPU step"""
        key = (self.name, self.max_subtokens_sequence_length, self.stride, self.maximum_subtoken_length, len(sentence))
        cache = getattr(sentence, "_kbner_tok", None)
        if cache is not None and cache[0] == key:
            return cache[1]
        out = self._tokenize_sentence(sentence)
        try:
            sentence._kbner_tok = (key, out)
        except AttributeError:
            pass
        return out

    def _tokenize_sentence(self, sentence):
        """-> (encoder rows [ids incl. <s>/</s>] -- one per sliding window --, window of each word token's first sub-token,
        its position inside that row (-1 = no sub-token))"""
        words = [self._eos_text if t.text == "<EOS>" else t.text for t in sentence]
        text = " ".join(words)
        pieces = self.tokenizer.tokenize(text)

        class _W:  # word tokens with <EOS> already substituted, as the reference deep-copies and rewrites them
            def __init__(self, s):
                self.text = s

        counts = self.reconstruct_tokens_from_subtokens([_W(w) for w in words], pieces)
        if any(c > self.maximum_subtoken_length for c in counts):
            kept, pos = [], 0
            for c in counts:
                kept += pieces[pos:pos + min(c, self.maximum_subtoken_length)]
                pos += c
            pieces = kept
            counts = [min(c, self.maximum_subtoken_length) for c in counts]
        ids = list(self.tokenizer.convert_tokens_to_ids(pieces))
        windows, where = self.split_windows(len(ids))
        first_row, first, g = [], [], 0
        for c in counts:
            if c > 0:
                w, pos = where(g)
                first_row.append(w)
                first.append(pos)
            else:
                first_row.append(0)
                first.append(-1)
            g += c
        rows = [[self._bos_id] + ids[lo:hi] + [self._eos_id] for lo, hi in windows]
        return rows, first_row, first

    def split_windows(self, n_ids):
        """Sliding windows over a sentence of n_ids content sub-tokens (flair/embeddings.py:3203-3227: encode_plus with
        max_length = max_subtokens_sequence_length, stride, return_overflowing_tokens -- every window holds up to
        max_length-2 content ids and the next one restarts `stride` ids before the previous one's end) and the seam rule
        of :3292-3299 (drop the last 1 + stride//2 positions of the accumulated states and the first 1 + stride//2 of the
        next window, then index the concatenation).  Returns ([(lo, hi) content range per window], where) with
        where(g) -> (window, position inside the window's row incl. <s>) for content position g."""
        W = self.max_subtokens_sequence_length - 2
        st = self.stride
        off = self.begin_offset
        if n_ids <= W:
            return [(0, n_ids)], (lambda g: (0, g + off))
        if not self.allow_long_sentences or st <= 0:
            return [(0, W)], (lambda g: (0, g + off) if g < W else (0, -1))  # truncated: no window covers g
        if st % 2 or st >= W:
            raise NotImplementedError("sliding window needs an even stride < max_subtokens_sequence_length - 2")
        starts = [0]
        while starts[-1] + W < n_ids:
            starts.append(starts[-1] + W - st)
        windows = [(lo, min(lo + W, n_ids)) for lo in starts]
        half = st // 2

        def where(g):
            for w, lo in enumerate(starts):
                if w == len(starts) - 1 or g < lo + W - half:
                    return w, g - lo + off
            raise AssertionError

        return windows, where

    # ------------------------------------------------------------------ document windows (v2_doc)
    def _doc_pieces(self, sentence):
        """(sub-token ids, per-token counts) of one sentence of a document, cached on the Sentence like the reference caches
        `subtoken_ids_sentence` / `token_subtoken_lengths` (embeddings.py:3700-3737); the v2 path tokenises the plain
        tokenised string -- no <EOS> substitution, no maximum_subtoken_length clamp"""
        cache = sentence.__dict__.setdefault("_kbner_doc", {})
        hit = cache.get(self.name)
        if hit is None:
            pieces = self.tokenizer.tokenize(sentence.to_tokenized_string())
            counts = self.reconstruct_tokens_from_subtokens(sentence.tokens, pieces) if pieces else [0] * len(sentence)
            hit = cache[self.name] = (list(self.tokenizer.convert_tokens_to_ids(pieces)), counts)
        return hit

    def _v2_window(self, sentence, max_sequence_length):
        """add_document_embeddings_v2 (embeddings.py:3657-3812): the sentence sits inside its document's sub-token stream
        (`sentence.doc`, position `sentence.doc_pos`); up to max_sequence_length ids are cut around it -- half of the free room
        on each side, the shorter side's slack going to the other -- and framed [CLS] .. [SEP].
        -> (row ids, position of the sentence's first sub-token inside the row, its per-token counts)"""
        stream, start, end = [], None, None
        counts = None
        for pos, ds in enumerate(sentence.doc):
            ids, cnt = self._doc_pieces(ds)
            if pos == sentence.doc_pos:
                start, counts = len(stream), cnt
            stream += ids
            if pos == sentence.doc_pos:
                end = len(stream)
        left, right, slen = start, len(stream) - end, end - start
        half = int((max_sequence_length - slen) / 2)
        if left < right:
            lc = min(left, half)
            rc = min(right, max_sequence_length - lc - slen)
        else:
            rc = min(right, half)
            lc = min(left, max_sequence_length - rc - slen)
        off = start - lc
        cls_id = self.tokenizer.convert_tokens_to_ids(self.tokenizer.cls_token)
        sep_id = self.tokenizer.convert_tokens_to_ids(self.tokenizer.sep_token)
        return [cls_id] + stream[off:end + rc] + [sep_id], start - off + 1, counts

    def _prepare_batch_v2doc(self, sentences):
        model_max = min(int(getattr(self.tokenizer, "model_max_length", 512) or 512) - 2, 510)
        B, n = len(sentences), max(len(s) for s in sentences)
        rows, firsts = [], []
        for s in sentences:
            if not hasattr(s, "doc") or len(self._doc_pieces(s)[0]) > self.max_subtokens_sequence_length:
                raise NotImplementedError("v2_doc: every sentence needs `.doc` / `.doc_pos` (ModelFinetuner(assign_doc_id=True, "
                                          "train_with_doc=True)) and must fit one window; over-long single sentences take the "
                                          "sliding-window path with v2_doc off")
            row, pos, counts = self._v2_window(s, model_max)
            f = []
            for c in counts:
                f.append(pos if c > 0 else -1)
                pos += c
            rows.append(row)
            firsts.append(f)
        S0 = max(len(r) for r in rows)
        ids = np.zeros((B, S0), np.int64)
        am = np.zeros((B, S0), np.int64)
        first = np.full((B, n), -1, np.int64)
        for b, (r, f) in enumerate(zip(rows, firsts)):
            ids[b, :len(r)] = r
            am[b, :len(r)] = 1
            first[b, :len(f)] = f
        return ids, am, first, np.asarray([len(s) for s in sentences], np.int64), np.tile(np.arange(B)[:, None], (1, n))

    def prepare_batch(self, sentences):
        """numpy integer batch: input_ids / attention_mask [R,S0] (R >= B encoder rows: a sentence longer than one window
        contributes several; padded with 0 like the reference), first_idx [B,n] (position inside the row, -1 pad),
        lengths [B], first_row [B,n] (which encoder row holds each word token's first sub-token)."""
        if getattr(self, "v2_doc", False):
            return self._prepare_batch_v2doc(sentences)
        toks = [self.tokenize_sentence(s) for s in sentences]
        B = len(toks)
        R = sum(len(t[0]) for t in toks)
        S0 = max(len(r) for t in toks for r in t[0])
        n = max(len(s) for s in sentences)
        ids = np.zeros((R, S0), np.int64)
        am = np.zeros((R, S0), np.int64)
        first = np.full((B, n), -1, np.int64)
        first_row = np.zeros((B, n), np.int64)
        lengths = np.zeros(B, np.int64)
        r0 = 0
        for b, (rows, fr, f) in enumerate(toks):
            for j, r in enumerate(rows):
                ids[r0 + j, :len(r)] = r
                am[r0 + j, :len(r)] = 1
            first[b, :len(f)] = f
            first_row[b, :len(f)] = np.asarray(fr, np.int64) + r0
            lengths[b] = len(sentences[b])
            r0 += len(rows)
        return ids, am, first, lengths, first_row

    def prepare_stack_batch(self, sentences):
        """integer batch for the frozen-stack path (BASELINE config 5).  With `use_internal_doc` the encoder reads each
        sentence's UNCHUNKED copy `sentence.doc_sent` (the sentence plus its retrieved context, embeddings.py:3116-3117) while
        only the tokens of the chunked sentence are pooled (:3283-3284: zip over the input sentence truncates)."""
        src = [getattr(s, "doc_sent", s) for s in sentences] if self.use_internal_doc else list(sentences)
        ids, am, first, lengths, first_row = self.prepare_batch(src)
        n = max(len(s) for s in sentences)
        B = len(sentences)
        f2 = np.full((B, n), -1, np.int64)
        r2 = np.zeros((B, n), np.int64)
        for b, s in enumerate(sentences):
            k = min(len(s), first.shape[1])
            f2[b, :k] = first[b, :k]
            r2[b, :k] = first_row[b, :k]
        return ids, am, f2, np.asarray([len(s) for s in sentences], np.int64), r2

    def _add_embeddings_internal(self, sentences):
        """stores the integer batch on the BatchedData (or returns it); the tagger's engine turns it into features"""
        batch = self.prepare_batch(sentences)
        if hasattr(sentences, "features"):
            sentences.features[self.name] = batch
        return sentences

    def train(self, mode=True):
        self.training = bool(mode) and self.fine_tune
        return self

    def extra_repr(self):
        return "model=%s" % self.name

    def __str__(self):
        return self.name


class BertEmbeddings(TokenEmbeddings):
    """BERT word embeddings (e.g. multilingual BERT in the ACE embedding pool), frozen, at inference.

    Behavioural reference (restated): flair/embeddings.py BertEmbeddings (:2667-2905): every word token is word-piece tokenised
    ON ITS OWN (:2733-2736), the pieces are framed [CLS] ... [SEP] and cut to max_sequence_length (:2737-2749), ids and mask are
    zero-padded (:2755-2758), the model runs with absolute positions and token type 0, and a token's embedding is the
    concatenation over `layers` (default the last four, '-1,-2,-3,-4') of the hidden state of its FIRST piece (:2861-2867;
    index = 1 + pieces of all earlier tokens -- also for a token that received no piece, which therefore borrows the next
    token's first piece).  `bert_model_or_path` must be a local directory (tokenizer files, config.json, weights)."""

    def __init__(self, bert_model_or_path: str = "bert-base-uncased", layers: str = "-1,-2,-3,-4", pooling_operation: str = "first",
                 use_scalar_mix: bool = False, fine_tune: bool = False, sentence_feat: bool = False, max_sequence_length=510):
        super().__init__()
        if not os.path.isdir(str(bert_model_or_path)):
            raise FileNotFoundError("BertEmbeddings(%r): a local model directory is required (no network)" % (bert_model_or_path,))
        if pooling_operation != "first" or use_scalar_mix or fine_tune or sentence_feat:
            raise NotImplementedError("BertEmbeddings: only frozen, pooling_operation='first', no scalar mix is on the MI355X path")
        from transformers import AutoTokenizer
        self.tokenizer = AutoTokenizer.from_pretrained(str(bert_model_or_path))
        config, sd = _load_hf_dir(str(bert_model_or_path))
        self.model = _EncoderHandle(config, sd)
        self.layer_indexes = [int(x) for x in str(layers).split(",")]
        if any(li >= 0 or -li > config.num_hidden_layers + 1 for li in self.layer_indexes):
            raise ValueError("layers must be negative indexes into the %d hidden states" % (config.num_hidden_layers + 1))
        self.pooling_operation = pooling_operation
        self.use_scalar_mix = False
        self.name = str(bert_model_or_path)
        self.fine_tune = False
        self.static_embeddings = True
        self.sentence_feat = False
        self.max_sequence_length = int(max_sequence_length)
        self._hidden = int(config.hidden_size)
        self._cls = self.tokenizer.convert_tokens_to_ids(self.tokenizer.cls_token)
        self._sep = self.tokenizer.convert_tokens_to_ids(self.tokenizer.sep_token)

    @property
    def embedding_length(self) -> int:
        return len(self.layer_indexes) * self._hidden

    def prepare_stack_batch(self, sentences):
        """-> ids, mask [B, S0], first [B, n] (position of each token's first piece, -1 padding), lengths [B]"""
        cache = self.__dict__.setdefault("_piece_cache", {})
        rows, counts = [], []
        for s in sentences:
            pieces, cnt = [], []
            for tok in s:
                p = cache.get(tok.text)
                if p is None:
                    p = cache[tok.text] = self.tokenizer.convert_tokens_to_ids(self.tokenizer.tokenize(tok.text))
                pieces.extend(p)
                cnt.append(len(p))
            rows.append(pieces)
            counts.append(cnt)
        longest = min(max(len(self.tokenizer.tokenize(s.to_tokenized_string())) for s in sentences), self.max_sequence_length)
        S0 = longest + 2
        B, n = len(sentences), max(len(s) for s in sentences)
        ids = np.zeros((B, S0), np.int64)
        am = np.zeros((B, S0), np.int64)
        first = np.full((B, n), -1, np.int64)
        for b, (pieces, cnt) in enumerate(zip(rows, counts)):
            pieces = pieces[:longest]
            row = [self._cls] + list(pieces) + [self._sep]
            ids[b, :len(row)] = row
            am[b, :len(row)] = 1
            pos = 1
            for k, c in enumerate(cnt):
                if pos >= S0:
                    raise IndexError("token %d of sentence %d lies beyond max_sequence_length (the reference fails here too)" % (k, b))
                first[b, k] = pos
                pos += c
        return ids, am, first, np.asarray([len(s) for s in sentences], np.int64)

    def _add_embeddings_internal(self, sentences):
        return sentences   # features are produced on the device by the tagger's stack engine

    def __str__(self):
        return self.name


class FlairEmbeddings(TokenEmbeddings):
    """Contextual string embeddings (Akbik et al. 2018) at inference: a character LM reads the whole tokenised sentence and
    the hidden state after each token's last character (forward LM) / before its first character read backwards (backward LM)
    is that token's embedding.

    Behavioural reference (restated): flair/embeddings.py FlairEmbeddings (:2271-2543): sentence text = tokens joined by one
    blank; every string of the batch is framed as "\n" + text (reversed for a backward LM) + " " and blank-padded to the
    longest (:2493-2510); token offsets :2520-2540.  `model` must be a local LanguageModel file (LanguageModel.save format):
    the named models ('en-forward', 'multi-backward', ...) are downloads, unavailable offline.  Only this class's bookkeeping
    is host code; the LM itself runs in kbner.stack.CharLM (HIP LSTM kernel)."""

    def __init__(self, model, fine_tune: bool = False, chars_per_chunk: int = 512, embedding_name: str = None):
        super().__init__()
        if fine_tune:
            raise NotImplementedError("fine-tuning a character LM is outside the hot path (FlairEmbeddings are frozen in config 5)")
        from flair.models.language_model import LanguageModel
        if isinstance(model, LanguageModel):
            self.lm = model
            self.name = "Task-LSTM-%s-%s-%s" % (self.lm.hidden_size, self.lm.nlayers, self.lm.is_forward_lm)
        else:
            if not os.path.exists(str(model)):
                raise FileNotFoundError("FlairEmbeddings(model=%r): a local character-LM file is required -- the named models are "
                                        "downloads (no network)" % (model,))
            self.lm = LanguageModel.load_language_model(model)
            self.name = str(model)
        if embedding_name is not None:
            self.name = embedding_name
        self.fine_tune = False
        self.static_embeddings = True
        self.is_forward_lm = self.lm.is_forward_lm
        self.chars_per_chunk = chars_per_chunk
        self._engine = None

    @property
    def embedding_length(self) -> int:
        return int(self.lm.hidden_size)

    def char_batch(self, sentences, n):
        """-> (char_ids int32 [steps, B], out_rows int32 [steps, B]): the framed / padded character ids and, per step, the
        token-major feature row (b * n + token) that receives the hidden state produced at that step, else -1"""
        texts = [s.to_tokenized_string() for s in sentences]
        longest = max(len(t) for t in texts)
        B = len(sentences)
        steps = longest + 2
        ids = np.zeros((steps, B), np.int32)
        rows = np.full((steps, B), -1, np.int32)
        get = self.lm.dictionary.get_idx_for_item
        blank = get(" ")
        for b, (s, t) in enumerate(zip(sentences, texts)):
            framed = "\n" + (t if self.is_forward_lm else t[::-1]) + " " + " " * (longest - len(t))
            ids[:, b] = [get(ch) for ch in framed] if framed else blank
            off_f, off_b = 1, len(t) + 1
            for k, tok in enumerate(s.tokens):
                off_f += len(tok.text)
                off = off_f if self.is_forward_lm else off_b
                rows[off, b] = b * n + k
                off_f += 1
                off_b -= 1 + len(tok.text)
        return ids, rows

    def engine(self, device):
        if self._engine is None:
            from kbner.stack import CharLM
            self._engine = CharLM(self.lm.state_dict(), self.lm.hidden_size, device)
        return self._engine

    def _add_embeddings_internal(self, sentences):
        return sentences   # features are produced on the device by the tagger's stack engine

    def __str__(self):
        return self.name


class _NeedsExternalWeights(TokenEmbeddings):
    """Embedding classes of the ACE config (config/xlmr-task-wiki-extdoc...ner6.yaml) whose weights are files this offline
    build cannot obtain ("ELMo/FastWord need external weight files and can stay stubbed").  They are
    constructible as PLACEHOLDERS: name and embedding_length as in the reference -- which is all the stacked tagger needs of an
    embedding its controller has deselected (`features * selection[idx]` with selection 0, sequence_tagger_model.py:889: the
    block of the concatenated input is zero whatever the embedding would have produced, and the BiLSTM's input weights keep the
    trained layout) -- and refuse, with an explanation, the moment they are asked for actual vectors."""
    _what = ""

    def _placeholder(self, name, length):
        self.name = name
        self.static_embeddings = True
        self._length = None if length is None else int(length)
        if self._length is None:
            self._refuse()
        log.warning("%s %r: %s are not available offline -- built as a placeholder of width %d that can only be used deselected "
                    "(its entry of `student.selection` / best_action must be 0)", type(self).__name__, name, self._what, self._length)

    @property
    def embedding_length(self) -> int:
        return self._length

    def _refuse(self):
        raise NotImplementedError("%s needs %s, which are not available offline; it can only take part deselected (its entry of "
                                  "`student.selection` 0), or drop it from the YAML's `embeddings:` block" %
                                  (type(self).__name__, self._what))

    def embed(self, sentences, *a, **k):
        self._refuse()

    def _add_embeddings_internal(self, sentences):
        self._refuse()


class ELMoEmbeddings(_NeedsExternalWeights):
    """reference: flair/embeddings.py:1214-1290 ELMoEmbeddings (allennlp ElmoEmbedder over options_file / weight_file); name
    'elmo-<model>', width = 3 layers x 2 directions x the projection size (3072 for the original / 5.5B models)"""
    _what = "the allennlp package and its ELMo options/weight files"

    def __init__(self, model: str = "original", options_file: str = None, weight_file: str = None, embedding_name=None,
                 is_hit_elmo=False, use_avg=False, embedding_length: int = None, **kwargs):
        super().__init__()
        import re
        proj = {"small": 128, "medium": 256, "original": 512, "large": 512, "pt": 512, "portuguese": 512, "pubmed": 512}.get(model)
        m = re.search(r"_(\d+)_\d+cnn", options_file or "")
        if m:
            proj = int(m.group(1))
        self._placeholder(embedding_name if embedding_name is not None else "elmo-" + model,
                          embedding_length if embedding_length is not None else (6 * proj if proj else None))


class FastWordEmbeddings(_NeedsExternalWeights):
    """reference: flair/embeddings.py:416-560 FastWordEmbeddings (gensim KeyedVectors lookup table, `freeze`); name
    'Word: <embeddings>', width = the vectors' size (300 for the fastText tables, 100 glove / twitter, 50 turian)"""
    _what = "the gensim word-vector files"

    def __init__(self, embeddings: str = None, all_tokens=None, field: str = None, if_cased: bool = True, freeze: bool = False,
                 additional_empty_embedding: bool = False, keepall: bool = False, embedding_name: str = None,
                 embedding_length: int = None, **kwargs):
        super().__init__()
        key = (embeddings or "").lower()
        width = {"glove": 100, "twitter": 100, "turian": 50, "extvec": 300}.get(key)
        if width is None and (key in ("crawl", "news", "en") or len(key) == 2 or key.endswith(("-crawl", "-news", "-wiki"))):
            width = 300     # fastText
        self._placeholder(embedding_name if embedding_name is not None else "Word: %s" % embeddings,
                          embedding_length if embedding_length is not None else width)
