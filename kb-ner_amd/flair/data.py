"""Sentence / Token / Label / Span / Dictionary / Corpus -- the host-side data model the tagger path uses.

Behavioural reference (restated, not copied): flair/data.py of KB-NER -- Dictionary (:21-112, byte-string keys and the
pickled {'idx2item','item2idx'} file format of resources/taggers/*.pkl), Label (:115), Token (:164), Span (:256),
Sentence (:340; get_spans :455-532, to_tokenized_string :615, convert_tag_scheme :630), Corpus (:837;
make_tag_dictionary :1083), iob2 / iob_iobes (:1122-1164)."""
import pickle
from collections import Counter, defaultdict
from typing import Dict, List, Optional


class Dictionary:
    """string <-> id map; items are stored as utf-8 bytes exactly like the reference's pickles."""

    def __init__(self, add_unk: bool = True):
        self.item2idx: Dict[bytes, int] = {}
        self.idx2item: List[bytes] = []
        self.multi_label = False
        if add_unk:
            self.add_item("<unk>")

    def add_item(self, item: str) -> int:
        key = item.encode("utf-8")
        idx = self.item2idx.get(key)
        if idx is None:
            idx = len(self.idx2item)
            self.idx2item.append(key)
            self.item2idx[key] = idx
        return idx

    def get_idx_for_item(self, item: str) -> int:
        return self.item2idx.get(item.encode("utf-8"), 0)

    def get_item_for_index(self, idx: int) -> str:
        return self.idx2item[idx].decode("utf-8")

    def get_items(self) -> List[str]:
        return [b.decode("utf-8") for b in self.idx2item]

    def __len__(self) -> int:
        return len(self.idx2item)

    def save(self, savefile):
        with open(savefile, "wb") as f:
            pickle.dump({"idx2item": self.idx2item, "item2idx": self.item2idx}, f)

    @classmethod
    def load_from_file(cls, filename: str) -> "Dictionary":
        d = cls(add_unk=False)
        with open(filename, "rb") as f:
            m = pickle.load(f, encoding="latin1")
        d.idx2item, d.item2idx = m["idx2item"], m["item2idx"]
        return d

    @classmethod
    def load(cls, name: str) -> "Dictionary":
        return cls.load_from_file(name)


class Label:
    def __init__(self, value: str, score: float = 1.0):
        self.value = value
        self.score = score

    @property
    def value(self):
        return self._value

    @value.setter
    def value(self, v):
        if v is None:
            raise ValueError("a label needs a value")
        self._value = v

    @property
    def score(self):
        return self._score

    @score.setter
    def score(self, s):
        self._score = s if 0.0 <= s <= 1.0 else 1.0   # the value is kept as given (an int 1 prints "1", data.py:132-137)

    def to_dict(self):
        return {"value": self.value, "confidence": self.score}

    def __repr__(self):
        return "%s (%s)" % (self._value, self._score)

    __str__ = __repr__


class Token:
    def __init__(self, text: str, idx: Optional[int] = None, head_id: Optional[int] = None,
                 whitespace_after: bool = True, start_position: Optional[int] = None):
        self.text = text
        self.idx = idx
        self.head_id = head_id
        self.whitespace_after = whitespace_after
        self.start_pos = start_position
        self.end_pos = start_position + len(text) if start_position is not None else None
        self.sentence = None
        self._embeddings: Dict = {}
        self.tags: Dict[str, Label] = {}

    def add_tag(self, tag_type: str, tag_value, confidence: float = 1.0):
        self.tags[tag_type] = tag_value if isinstance(tag_value, Label) else Label(tag_value, confidence)

    add_tag_label = add_tag

    def get_tag(self, tag_type: str) -> Label:
        return self.tags.get(tag_type) or Label("")

    def get_tags_proba_dist(self, tag_type: str):
        return []

    def set_embedding(self, name, vector):
        self._embeddings[name] = vector

    def clear_embeddings(self, embedding_names=None):
        if embedding_names is None:
            self._embeddings = {}
        else:
            for n in embedding_names:
                self._embeddings.pop(n, None)

    def to(self, device):
        pass

    @property
    def embedding(self):
        return self.get_embedding()

    def get_embedding(self):
        import torch
        vs = [self._embeddings[k] for k in sorted(self._embeddings)]
        return torch.cat(vs, dim=0) if vs else torch.zeros(0)

    def __repr__(self):
        return "Token: %s %s" % (self.idx, self.text)

    __str__ = __repr__


class Span:
    def __init__(self, tokens: List[Token], tag: Optional[str] = None, score: float = 1.0):
        self.tokens = tokens
        self.tag = tag
        self.score = score
        self.start_pos = tokens[0].start_pos if tokens else None
        self.end_pos = tokens[-1].end_pos if tokens else None

    @property
    def text(self) -> str:
        return " ".join(t.text for t in self.tokens)

    def to_original_text(self) -> str:
        out = ""
        for t in self.tokens:
            out += t.text + (" " if t.whitespace_after else "")
        return out.strip()

    def to_dict(self):
        return {"text": self.to_original_text(), "start_pos": self.start_pos, "end_pos": self.end_pos, "type": self.tag,
                "confidence": self.score}

    def __str__(self):
        ids = ",".join(str(t.idx) for t in self.tokens)
        return '%s-span [%s]: "%s"' % (self.tag, ids, self.text) if self.tag is not None else 'span [%s]: "%s"' % (ids, self.text)

    def __repr__(self):
        ids = ",".join(str(t.idx) for t in self.tokens)
        return '<%s-span (%s): "%s">' % (self.tag, ids, self.text) if self.tag is not None else '<span (%s): "%s">' % (ids, self.text)


_BIOES = ("B-", "I-", "O-", "E-", "S-")


def span_tables(items: List[str]):
    """per tag id of a tag dictionary: (prefix code, class id) under get_spans' normalisation ('' / 'O' -> outside, a bare class
    name -> a single-token span), + the class names.  Prefix codes: 0 O, 1 B, 2 I, 3 E, 4 S."""
    import numpy as np
    code = {"O-": 0, "B-": 1, "I-": 2, "E-": 3, "S-": 4}
    pref, cls, names = [], [], []
    for val in items:
        if val in ("", "O"):
            val = "O-"
        if val[:2] not in _BIOES:
            val = "S-" + val
        pref.append(code[val[:2]])
        c = val[2:]
        if c not in names:
            names.append(c)
        cls.append(names.index(c))
    return np.asarray(pref, np.int8), np.asarray(cls, np.int32), names


def batch_spans(sentences, ids, scores, tables, min_score: float = -1, skip_class: str = None, drop_flags=None):
    """Sentence.get_spans for a whole batch from TAG-ID arrays instead of per-token Label objects: ids int[B, n] (tag ids of
    `tables`' dictionary; columns >= len(sentence) ignored), scores float[B, n] or None (= 1.0).  Returns, per sentence, the
    same Span list get_spans(tag_type, min_score, skip_class, drop_touching) returns for tokens carrying those tags
    (drop_flags bool[B, n]: token positions whose presence drops a span).  The segmentation rule of flair/data.py:455-532 --
    a span is a maximal run of non-O tokens, cut before every B- / S- token and before a token that follows an S- of another
    class -- is evaluated with numpy over the flattened batch; Python only touches the spans that survive the filters, which
    is what makes evaluate() cheap on sentences that are 95 % single-token S-X context."""
    import numpy as np
    pref_t, cls_t, names = tables
    ids = np.asarray(ids)
    B, n = ids.shape
    lens = np.asarray([len(s) for s in sentences], np.int64)
    n1 = n + 1                                       # one outside column per row: spans never run across rows
    valid = np.zeros((B, n1), bool)
    valid[:, :n] = np.arange(n)[None, :] < lens[:, None]
    pref = np.zeros((B, n1), np.int8)
    cls = np.full((B, n1), -1, np.int32)
    pref[:, :n] = np.where(valid[:, :n], pref_t[ids], 0)
    cls[:, :n] = np.where(valid[:, :n], cls_t[ids], -1)
    inside = (pref != 0).ravel()
    pref, cls = pref.ravel(), cls.ravel()
    prev_pref = np.concatenate(([0], pref[:-1]))
    prev_cls = np.concatenate(([-1], cls[:-1]))
    prev_inside = np.concatenate(([False], inside[:-1]))
    opens = (pref == 1) | (pref == 4) | ((prev_pref == 4) & (prev_cls != cls) & inside)
    starts = inside & (opens | ~prev_inside)
    spos = np.flatnonzero(starts)
    out = [[] for _ in range(B)]
    if spos.size == 0:
        return out
    bpos = np.flatnonzero(starts | ~inside)          # where a running span ends (exclusive)
    epos = bpos[np.searchsorted(bpos, spos, side="right")]
    length = epos - spos
    keep = np.ones(spos.size, bool)
    if skip_class is not None and skip_class in names:
        keep &= ~((length == 1) & (cls[spos] == names.index(skip_class)))       # (longer spans: voted class, decided below)
    if drop_flags is not None:
        fl = np.zeros((B, n1), np.int64)
        fl[:, :n] = np.asarray(drop_flags, bool)
        cs = np.concatenate(([0], np.cumsum(fl.ravel())))
        keep &= (cs[epos] - cs[spos]) == 0
    if scores is not None:
        scores = np.asarray(scores)
    for s0, e0 in zip(spos[keep].tolist(), epos[keep].tolist()):
        b, i0 = divmod(s0, n1)
        i1 = i0 + (e0 - s0)
        if e0 - s0 == 1:
            best = names[cls[s0]]
        else:
            votes: Dict[str, float] = defaultdict(float)
            for j in range(s0, e0):
                votes[names[cls[j]]] += 1.1 if opens[j] else 1.0
            best = sorted(votes.items(), key=lambda kv: kv[1], reverse=True)[0][0]
            if skip_class is not None and best == skip_class:
                continue
        mean = 1.0
        if scores is not None:
            # the same float arithmetic as get_spans: python sum of the token scores / count
            vals = [float(x) for x in scores[b, i0:i1]]
            mean = sum(vals) / len(vals)
        if mean > min_score:
            out[b].append(Span(sentences[b].tokens[i0:i1], tag=best, score=mean))
    return out


class Sentence:
    """A list of Tokens (+ sentence-level labels).  `Sentence("a b c")` splits on whitespace."""

    def __init__(self, text: Optional[str] = None, use_tokenizer: bool = False, labels=None, language_code: Optional[str] = None):
        self.tokens: List[Token] = []
        self.labels: List[Label] = []
        if labels is not None:
            self.add_labels(labels)
        self._embeddings: Dict = {}
        self.language_code = language_code
        self.tokenized = None
        # knowledge-distillation targets, one entry per teacher (flair/data.py:364-370 of the reference)
        self._teacher_target, self._teacher_weights, self._teacher_posteriors = [], [], []
        self._teacher_startscores, self._teacher_endscores = [], []
        self._teacher_prediction = []
        if text is not None:
            pos = 0
            for word in text.split():
                start = text.index(word, pos)
                self.add_token(Token(word, start_position=start))
                pos = start + len(word)

    # -- container protocol
    def __len__(self):
        return len(self.tokens)

    def __iter__(self):
        return iter(self.tokens)

    def __getitem__(self, i):
        return self.tokens[i]

    def get_token(self, token_id: int) -> Optional[Token]:
        for t in self.tokens:
            if t.idx == token_id:
                return t
        return None

    def add_token(self, token):
        if isinstance(token, str):
            token = Token(token)
        token.sentence = self
        if token.idx is None:
            token.idx = len(self.tokens) + 1
        self.tokens.append(token)

    def add_label(self, label):
        self.labels.append(label if isinstance(label, Label) else Label(label))

    def add_labels(self, labels):
        for lab in labels:
            self.add_label(lab)

    def get_label_names(self):
        return [lab.value for lab in self.labels]

    # -- spans (flair/data.py:455-532)
    def get_spans(self, tag_type: str, min_score: float = -1, skip_class: str = None, drop_touching=None) -> List[Span]:
        """skip_class / drop_touching (not in the reference) are POST-FILTERS applied before a Span object is built, exactly
        equivalent to filtering the reference's list afterwards: spans whose class is `skip_class` and spans containing a
        token whose 1-based idx is in `drop_touching` are not returned.  evaluate()'s remove_x rule (sequence_tagger_model.py
        :2653-2672) drops the hundreds of single-token S-X context spans per sentence anyway; not materialising them is what
        keeps the host loop fast."""
        spans: List[Span] = []
        cur: List[Token] = []
        votes: Dict[str, float] = defaultdict(float)

        def close():
            nonlocal cur, votes
            if cur:
                sc = [t.get_tag(tag_type).score for t in cur]
                mean = sum(sc) / len(sc)
                if mean > min_score:
                    best = sorted(votes.items(), key=lambda kv: kv[1], reverse=True)[0][0]
                    if not (skip_class is not None and best == skip_class) and \
                            not (drop_touching and any(t.idx in drop_touching for t in cur)):
                        spans.append(Span(cur, tag=best, score=mean))
            cur, votes = [], defaultdict(float)

        prev = "O"
        for tok in self.tokens:
            val = tok.get_tag(tag_type).value
            if val in ("", "O"):
                val = "O-"
            if val[:2] not in _BIOES:
                val = "S-" + val           # a bare class name counts as a single-token span
            inside = val[:2] != "O-"
            opens = val[:2] in ("B-", "S-")
            if prev[:2] == "S-" and prev[2:] != val[2:] and inside:
                opens = True
            if (opens or not inside) and cur:
                close()
            if inside:
                cur.append(tok)
                votes[val[2:]] += 1.1 if opens else 1.0
            prev = val
        close()
        return spans

    def chunk_sentence(self, start_idx: int, end_idx: int):
        """keep tokens [start_idx, end_idx) and renumber them (data.py:704-715); used to cut a sentence at its <EOS> token"""
        kept = []
        for i in range(start_idx, end_idx):
            tok = self.tokens[i]
            kept.append(tok)
            tok.idx = len(kept)
        self.tokens = kept
        self.tokenized = " ".join(t.text for t in self.tokens)
        if hasattr(self, "_kbner_tok"):
            del self._kbner_tok

    # -- strings
    def to_tokenized_string(self) -> str:
        if self.tokenized is None:
            self.tokenized = " ".join(t.text for t in self.tokens)
        return self.tokenized

    def to_plain_string(self) -> str:
        out = ""
        for t in self.tokens:
            out += t.text + (" " if t.whitespace_after else "")
        return out.rstrip()

    to_original_text = to_plain_string

    def to_tagged_string(self, main_tag=None) -> str:
        parts = []
        for t in self.tokens:
            parts.append(t.text)
            tg = [lab.value for k, lab in t.tags.items() if (main_tag is None or k == main_tag) and lab.value not in ("", "O")]
            if tg:
                parts.append("<%s>" % "/".join(tg))
        return " ".join(parts)

    def convert_tag_scheme(self, tag_type: str = "ner", target_scheme: str = "iob"):
        tags = [t.get_tag(tag_type) for t in self.tokens]
        if target_scheme in ("iob", "iobes"):
            iob2(tags)
        if target_scheme == "iobes":
            tags = iob_iobes(tags)
        for t, tg in zip(self.tokens, tags):
            t.add_tag(tag_type, tg)

    def infer_space_after(self):
        """CoNLL-style files carry no spacing; guess it (quotes and closing punctuation attach to the left)."""
        last, quotes = None, 0
        for t in self.tokens:
            if t.text == '"':
                quotes += 1
                if quotes % 2 != 0:
                    t.whitespace_after = False
                elif last is not None:
                    last.whitespace_after = False
            if last is not None:
                if t.text in (".", ":", ",", ";", ")", "n't", "!", "?"):
                    last.whitespace_after = False
                if t.text.startswith("'"):
                    last.whitespace_after = False
            if t.text == "(":
                t.whitespace_after = False
            last = t
        return self

    # -- embeddings bookkeeping
    def set_embedding(self, name, vector):
        self._embeddings[name] = vector

    def clear_embeddings(self, embedding_names=None, also_clear_word_embeddings: bool = True):
        self._embeddings = {} if embedding_names is None else {k: v for k, v in self._embeddings.items() if k not in embedding_names}
        if also_clear_word_embeddings:
            for t in self.tokens:
                t.clear_embeddings(embedding_names)

    def to(self, device):
        pass

    # -- teacher targets of `distill_mode` training (reference flair/data.py:762-806).  The reference stores torch tensors padded
    # to the length of the teacher's batch and moves them with `storage_mode`; here they are host numpy arrays trimmed to the
    # sentence (rows past the sentence length are zero in the reference and never read), padded per training batch by
    # FastSequenceTagger._kd_batch.  `storage_mode` is accepted and ignored.
    @staticmethod
    def _host(vector):
        import numpy as np
        return vector.detach().cpu().numpy() if hasattr(vector, "detach") else np.asarray(vector)

    def set_teacher_target(self, vector, storage_mode=None):       # int [len, best_k] n-best tag sequences
        self._teacher_target.append(self._host(vector)[:len(self)])

    def set_teacher_prediction(self, vector, storage_mode=None):   # f32 [len, T] teacher emissions (softmax of them: distill_prob)
        self._teacher_prediction.append(self._host(vector)[:len(self)])

    def set_teacher_weights(self, vector, storage_mode=None):      # f32 [best_k] path weights
        self._teacher_weights.append(self._host(vector))

    def set_teacher_posteriors(self, vector, storage_mode=None):   # f32 [len, T] fb scores, or [len - 1, T * T] (distill_exact)
        self._teacher_posteriors.append(self._host(vector))

    def set_teacher_startscores(self, vector, storage_mode=None):  # f32 [T]
        self._teacher_startscores.append(self._host(vector))

    def set_teacher_endscores(self, vector, storage_mode=None):    # f32 [T]
        self._teacher_endscores.append(self._host(vector))

    def get_teacher_target(self):
        import numpy as np
        return np.concatenate(self._teacher_target, -1)

    def get_teacher_prediction(self, pooling="mean", weight=None):
        """flair/data.py:786-807, the pooling a CRF student uses: the mean over the teachers"""
        import numpy as np
        if pooling != "mean" or weight is not None:
            raise NotImplementedError("weighted teacher pooling (language attention) is outside the hot path")
        return np.stack(self._teacher_prediction).mean(0)

    def get_teacher_weights(self):
        import numpy as np
        return np.concatenate(self._teacher_weights, -1)

    def get_teacher_posteriors(self):
        import numpy as np
        return np.stack(self._teacher_posteriors, -2)

    def get_teacher_startscores(self):
        import numpy as np
        return np.stack(self._teacher_startscores, -2)

    def get_teacher_endscores(self):
        import numpy as np
        return np.stack(self._teacher_endscores, -2)

    def get_language_code(self) -> str:
        return self.language_code or "en"

    def to_dict(self, tag_type: str = None):
        d = {"text": self.to_original_text(), "labels": [lab.to_dict() for lab in self.labels]}
        if tag_type:
            d["entities"] = [s.to_dict() for s in self.get_spans(tag_type)]
        return d

    def __repr__(self):
        return 'Sentence: "%s" - %d Tokens' % (" ".join(t.text for t in self.tokens), len(self.tokens))

    __str__ = __repr__


def iob2(tags: List[Label]) -> bool:
    """In-place IOB1 -> IOB2 (an I- that opens a chunk becomes B-); False on a malformed tag."""
    for i, tag in enumerate(tags):
        if tag.value == "O":
            continue
        parts = tag.value.split("-")
        if len(parts) != 2 or parts[0] not in ("I", "B"):
            return False
        if parts[0] == "B":
            continue
        if i == 0 or tags[i - 1].value == "O" or tags[i - 1].value[1:] != tag.value[1:]:
            tags[i].value = "B" + tag.value[1:]
    return True


def iob_iobes(tags: List[Label]) -> List[str]:
    """IOB2 -> IOBES: a B not followed by I becomes S, an I not followed by I becomes E."""
    out = []
    n = len(tags)
    for i, tag in enumerate(tags):
        v = tag.value
        if v == "O":
            out.append(v)
            continue
        head = v.split("-")[0]
        nxt_is_i = i + 1 < n and tags[i + 1].value.split("-")[0] == "I"
        if head == "B":
            out.append(v if nxt_is_i else v.replace("B-", "S-"))
        elif head == "I":
            out.append(v if nxt_is_i else v.replace("I-", "E-"))
        else:
            raise Exception("Invalid IOB format!")
    return out


class Corpus:
    def __init__(self, train, dev, test, name: str = "corpus"):
        self._train, self._dev, self._test = train, dev, test
        self.name = name

    @property
    def train(self):
        return self._train

    @property
    def dev(self):
        return self._dev

    @property
    def test(self):
        return self._test

    def get_all_sentences(self):
        out = []
        for part in (self.train, self.dev, self.test):
            if part is not None:
                out.extend(part[i] for i in range(len(part)))
        return out

    def make_tag_dictionary(self, tag_type: str) -> Dictionary:
        """'O' first, then tags in corpus order, then <START>/<STOP> (flair/data.py:1083-1097)."""
        d = Dictionary()
        d.add_item("O")
        for s in self.get_all_sentences():
            for t in s.tokens:
                d.add_item(t.get_tag(tag_type).value)
        d.add_item("<START>")
        d.add_item("<STOP>")
        return d

    def make_vocab_dictionary(self, max_tokens=-1, min_freq=1) -> Dictionary:
        cnt = Counter(t.text for s in self.train for t in s.tokens)
        d = Dictionary()
        for tok, freq in cnt.most_common():
            if freq < min_freq or (max_tokens != -1 and len(d) >= max_tokens):
                break
            d.add_item(tok)
        return d

    def obtain_statistics(self, tag_type: str = None, pretty_print: bool = True):
        stats = {}
        for name, part in (("TRAIN", self.train), ("TEST", self.test), ("DEV", self.dev)):
            if part is None:
                continue
            lens = [len(part[i]) for i in range(len(part))]
            stats[name] = {"dataset": name, "total_number_of_documents": len(lens),
                           "number_of_tokens": {"total": sum(lens), "min": min(lens) if lens else 0,
                                                "max": max(lens) if lens else 0, "avg": sum(lens) / max(1, len(lens))}}
        if pretty_print:
            import json
            return json.dumps(stats, indent=4)
        return stats

    def __str__(self):
        return "Corpus: %d train + %d dev + %d test sentences" % (len(self.train), len(self.dev), len(self.test))


class MultiCorpus(Corpus):
    def __init__(self, corpora: List[Corpus], name: str = "multicorpus"):
        from torch.utils.data import ConcatDataset
        self.corpora = corpora
        super().__init__(ConcatDataset([c.train for c in corpora]), ConcatDataset([c.dev for c in corpora]),
                         ConcatDataset([c.test for c in corpora]), name=name)

    def __str__(self):
        return "\n".join(str(c) for c in self.corpora)


class FlairDataset:
    def is_in_memory(self) -> bool:
        return True
