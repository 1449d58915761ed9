#!/usr/bin/env python
"""TEST INFRASTRUCTURE (container only).  Golden knowledge-augmented CoNLL files produced by RUNNING the reference's
kb/context_process.py functions `process_google` (:213-502) and `write_file` (:660-672).  The file is a script whose module level
needs ElasticSearch dumps, private paths and a tokenizer download, so the two function definitions are taken from its `ast`
(read from /root/reference at generation time, executed in a scratch namespace, never copied) with the module globals they
read: `tokenizer` (the tiny local tokenizer behind the 3.0.0 adapter), `use_xlmr_tokenization = True` (:760).
    python oracle/gen_golden_context.py  -> tests/golden/context_format.json
Inputs: sentences as 4-column CoNLL lines + a retrieval dictionary {lower-cased sentence text: [contexts in rank order]} that
exercises the budget rule (a context that does not fit is skipped, a later shorter one is taken), non-printable characters,
a sentence with no retrieval hit, a sentence whose only context does not fit; train (max_len = length_limit) and dev / test
(max_len = 999) writers."""
import ast
import json
import os
import re
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden")
SRC = "/root/reference/kb/context_process.py"

from oracle import ref_import  # noqa: E402


def load_functions(names):
    tree = ast.parse(open(SRC, encoding="utf-8").read())
    picked = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            picked[node.name] = node      # later definitions of the same name win, as at import time
    mod = ast.Module(body=[picked[n] for n in names], type_ignores=[])
    ns = {"re": re, "np": np, "pdb": None, "os": os, "failed_id": 0}
    exec(compile(mod, SRC, "exec"), ns)
    return ns


def main():
    import tiny_assets
    tdir = tempfile.mkdtemp(prefix="ctx_")
    tok = ref_import.TokenizerAdapter(tiny_assets.build_tokenizer_dir(tdir))
    ns = load_functions(["process_google", "write_file"])
    ns["tokenizer"] = tok
    ns["use_xlmr_tokenization"] = True
    rng = np.random.default_rng(11)
    W = tiny_assets.WORDS

    def ctx(k):
        return " ".join(str(w) for w in rng.choice(W, size=k))

    sents = [["alice NNP PROPN B-PER", "visited VBD VERB O", "berlin NNP PROPN B-LOC"],
             ["zalando NNP PROPN B-CORP", "research NN NOUN I-CORP", "is VBZ AUX O", "in IN ADP O", "berlin NNP PROPN B-LOC"],
             ["bob NNP PROPN B-PER", "lives VBZ VERB O", "near IN ADP O", "paris NNP PROPN B-LOC"],
             ["carol NNP PROPN B-PER", "works VBZ VERB O", "at IN ADP O", "google NNP PROPN B-CORP"],
             ["the DT DET O", "quick JJ ADJ O", "brown JJ ADJ O", "fox NN NOUN O"]]
    gd = {"alice visited berlin": [ctx(9), ctx(12) + " \x07bell‎ mark", ctx(7)],
          "zalando research is in berlin": [ctx(10), ctx(60), ctx(6), ctx(5), ctx(30), ctx(4)],     # 60 / 30 do not fit a limit of 60
          "carol works at google": [ctx(80)],                                                       # the only context never fits
          "the quick brown fox": [ctx(8), ctx(8), ctx(8), ctx(8), ctx(8), ctx(8), ctx(8)]}           # stops when < 10 sub-tokens remain
    out = {"sentences": sents, "google_dict": gd, "runs": []}
    for limit, max_len in ((60, 60), (60, 999), (510, 510), (40, 40)):
        new, failed = ns["process_google"]([list(s) for s in sents], {k: list(v) for k, v in gd.items()}, [], is_conll=True,
                                           clean_file=False, full_doc=True, add_eos=True, length_limit=limit, for_luke=False,
                                           lang="en", is_wiki_retrieval=True)
        path = os.path.join(tdir, "out_%d_%d.txt" % (limit, max_len))
        ns["write_file"](path, new, max_len=max_len)
        out["runs"].append({"length_limit": limit, "max_len": max_len, "file": open(path, encoding="utf-8").read(),
                            "failed": failed, "subtokens": [len(tok.tokenize(re.sub("<EOS>", tok._eos_token, " ".join(w.split()[0] for w in s))))
                                                            for s in new]})
    with open(os.path.join(GOLD, "context_format.json"), "w") as f:
        json.dump(out, f, indent=1, ensure_ascii=False)
    print("wrote context_format.json", os.path.getsize(os.path.join(GOLD, "context_format.json")), "bytes;",
          [(r["length_limit"], r["file"].count("<EOS>"), r["subtokens"]) for r in out["runs"]])


if __name__ == "__main__":
    main()
