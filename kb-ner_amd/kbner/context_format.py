"""The on-disk format either side of the hot path: the knowledge-augmented CoNLL files that
kb/context_process.py writes and ColumnCorpus reads.  The retrieval itself (ElasticSearch over a Wikipedia dump) is
offline preprocessing and out of scope; what the tagger depends on is the FILE CONVENTION, restated here as a writer
(for synthetic / user-supplied contexts) and a validator:

    # id <anything>                      comment line (ColumnCorpus comment_symbol '# id')
    word POS UPOS NER                    the sentence proper (4 columns when is_conll, kb/context_process.py:218-221)
    <EOS> B-X B-X B-X                    separator, only if at least one context follows (:464-477)
    ctxword B-X B-X B-X                  retrieved context, every token labelled B-X (:424-426; S-X after IOBES conversion)
    <blank line>

Budget (:403-405, 428-436, 974): contexts are taken in rank order while the XLM-R sub-token count of
sentence + contexts + <EOS> stays <= length_limit (510 for train files); a context that does not fit is SKIPPED (later,
shorter ones may still fit); the scan stops once fewer than 10 sub-tokens remain."""
from typing import Callable, Iterable, List, Optional, Sequence, Tuple

EOS_LINE = "<EOS> B-X B-X B-X"
X_COLS = "B-X B-X B-X"


def select_contexts(n_sentence_subtokens: int, contexts: Sequence[str], count: Callable[[str], int], length_limit: int = 510,
                    add_eos: bool = True) -> List[str]:
    """the contexts (printable characters only) that kb/context_process.py:396-440 keeps, in order"""
    used, total = [], n_sentence_subtokens
    for cxt in contexts:
        if length_limit - total < 10:
            break
        cxt = "".join(c for c in cxt if c.isprintable())
        words = cxt.split()
        if not words:
            continue
        n = count(" ".join(words))
        if n + total + (1 if add_eos else 0) > length_limit:
            continue
        total += n
        used.append(" ".join(words))
    return used


def format_sentence(tokens: Sequence[Tuple[str, str, str, str]], contexts: Sequence[str], sent_id: Optional[str] = None) -> List[str]:
    """lines of one augmented sentence (without the trailing blank line)"""
    lines = ["# id %s" % sent_id] if sent_id is not None else []
    lines += ["%s %s %s %s" % t for t in tokens]
    ctx_words = [w for c in contexts for w in c.split()]
    if ctx_words:
        lines.append(EOS_LINE)
        lines += ["%s %s" % (w, X_COLS) for w in ctx_words]
    return lines


def write_file(path: str, sentences: Iterable[dict], count: Callable[[str], int], length_limit: int = 510,
               max_lines: Optional[int] = None) -> int:
    """sentences: dicts with 'tokens' [(word, pos, upos, ner)], 'contexts' [str, ...] in rank order, optional 'id'.
    `count(text)` = number of XLM-R sub-tokens of a space-joined string (tokenizer.tokenize).  max_lines: sentences with more
    lines (word tokens incl. <EOS> and context) are dropped, as kb/context_process.py's write_file(max_len) does -- it passes
    length_limit for train files and 999 for dev / test (:995-1000).  Returns #sentences written.
    The sentence's own sub-token count is taken on its lower-cased text (`keyword.lower()` for wiki retrieval, :296-306)."""
    n = 0
    with open(path, "w", encoding="utf-8") as f:
        for s in sentences:
            text = " ".join(t[0] for t in s["tokens"])
            used = select_contexts(count(text), s.get("contexts", ()), count, length_limit)
            lines = format_sentence(s["tokens"], used, s.get("id"))
            if max_lines is not None and len([ln for ln in lines if not ln.startswith("# id")]) > max_lines:
                continue
            f.write("\n".join(lines) + "\n\n")
            n += 1
    return n


class FormatError(ValueError):
    pass


def validate_file(path: str, count: Optional[Callable[[str], int]] = None, length_limit: Optional[int] = 510,
                  eos_text: str = "</s>", columns: int = 4) -> dict:
    """Checks the convention above; with `count`, also the sub-token budget the way :478-490 measures it (the literal
    <EOS> replaced by the tokenizer's eos string).  Returns statistics; raises FormatError naming file:line."""
    stats = dict(sentences=0, with_context=0, real_tokens=0, context_tokens=0, max_subtokens=0, over_budget=0)
    cur: List[Tuple[int, List[str]]] = []

    def flush():
        if not cur:
            return
        words = [c[1][0] for c in cur]
        ner = [c[1][-1] for c in cur]
        eos = [i for i, w in enumerate(words) if w == "<EOS>"]
        if len(eos) > 1:
            raise FormatError("%s:%d: more than one <EOS> in a sentence" % (path, cur[eos[1]][0]))
        k = eos[0] if eos else len(words)
        for i in range(k):
            if ner[i].endswith("-X"):
                raise FormatError("%s:%d: X label before <EOS>" % (path, cur[i][0]))
        for i in range(k, len(words)):
            if ner[i] != "B-X":
                raise FormatError("%s:%d: context token not labelled B-X" % (path, cur[i][0]))
        if eos and k == len(words) - 1:
            raise FormatError("%s:%d: <EOS> without context" % (path, cur[k][0]))
        stats["sentences"] += 1
        stats["with_context"] += 1 if eos else 0
        stats["real_tokens"] += k
        stats["context_tokens"] += max(0, len(words) - k - 1)
        if count is not None:
            n = count(" ".join(eos_text if w == "<EOS>" else w for w in words))
            stats["max_subtokens"] = max(stats["max_subtokens"], n)
            if length_limit is not None and n > length_limit:
                stats["over_budget"] += 1
        cur.clear()

    with open(path, encoding="utf-8") as f:
        for ln, line in enumerate(f, 1):
            line = line.rstrip("\n")
            if line.startswith("# id"):
                continue
            if not line.strip():
                flush()
                continue
            parts = line.split()
            if len(parts) != columns:
                raise FormatError("%s:%d: expected %d columns, got %d" % (path, ln, columns, len(parts)))
            cur.append((ln, parts))
    flush()
    return stats
