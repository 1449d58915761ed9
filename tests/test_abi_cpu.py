"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/kbner.h declares
(no compute calls without a GPU)."""
import os
import re

import numpy as np


def test_build_and_symbols():
    import __graft_entry__ as ge
    path = ge.build()
    assert os.path.exists(path)
    from kbner import lib
    handle = lib.load()
    hdr = open(os.path.join(ge.ROOT, "include", "kbner.h")).read()
    declared = set(re.findall(r"\b(kbner_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(handle, name), "missing export: " + name
    # the binding lists exactly the header's symbols
    assert declared == set(lib.SIGNATURES), (declared ^ set(lib.SIGNATURES))
    assert handle.kbner_abi_version() == 2


def test_integration_doc_maps_every_symbol():
    """INTEGRATION.md section C names every entry point of the header (what it replaces in the reference, who calls it here)."""
    import __graft_entry__ as ge
    hdr = open(os.path.join(ge.ROOT, "include", "kbner.h")).read()
    declared = set(re.findall(r"\b(kbner_[a-z0-9_]+)\s*\(", hdr))
    doc = open(os.path.join(ge.ROOT, "INTEGRATION.md")).read()
    table = doc[doc.index("## C. Symbol map"):]
    missing = sorted(n for n in declared if "`%s`" % n not in table)
    assert not missing, missing


def test_product_path_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under kb-ner_amd/ imports it (the checker behind smoke() lives in tests/)."""
    import __graft_entry__ as ge
    pkg = os.path.join(ge.PKG)
    bad = []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith(".py"):
                continue
            src = open(os.path.join(dirpath, f)).read()
            if re.search(r"^\s*(from|import)\s+(oracle|selftest)\b", src, re.M):
                bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_batch_assembly_matches_oracle_compaction():
    from kbner import batch as kb
    from oracle import crf as ocrf
    b = kb.synthetic_batch(3, 128, vocab=1000, n_real=5, seed=1)
    assert b["S"] == 128 and b["ids"].shape[0] == 512  # 3*128 rows padded up to a multiple of 256
    # position ids: cumsum over ids != pad, + pad
    ids = b["input_ids"]
    nz = (ids != 1).astype(np.int64)
    np.testing.assert_array_equal(b["pos_ids"][:384].reshape(3, 128), np.cumsum(nz, 1) * nz + 1)
    # compaction index agrees with the oracle's remove_x compaction on fake features
    n = b["first_idx"].shape[1]
    feats = np.random.default_rng(0).standard_normal((3, n, 29)).astype(np.float32)
    cf, ct, lens, keep = ocrf.compact_remove_x(feats, b["tags"], b["lengths"], 9)
    np.testing.assert_array_equal(lens, b["clens"])
    np.testing.assert_array_equal(keep, b["keep"])
    nc = b["ctags"].shape[1]
    np.testing.assert_array_equal(ct[:, :nc], b["ctags"][:, :ct.shape[1]])
    flat = b["row_idx"].reshape(3, n)
    for r in range(3):
        k = np.nonzero(keep[r])[0]
        np.testing.assert_array_equal(b["crow_idx"].reshape(3, nc)[r, :len(k)], flat[r, k])
