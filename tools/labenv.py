"""Lab switches of the tools (never imported by the product package; kb-ner_amd/kbner/lib.py reads no environment variable):
    KBNER_LIB            an experiment build of the library to load instead of the in-tree libkbner_hip.so
    KBNER_GEMM_VARIANT   kbner_gemm_set_variant(<int>) right after loading (include/kbner.h)
Call apply() before the first kbner.lib.load() / kbner.ops call."""
import os


def apply():
    import torch  # noqa: F401  (FIRST: torch brings its own HIP runtime; loading libkbner_hip.so before it binds the library to a second
    #                 copy of libamdhip64 that never sees torch's device context -- every launch then fails with hipErrorNoDevice)
    from kbner import lib
    p = os.environ.get("KBNER_LIB")
    if p:
        if lib._lib is not None:
            raise RuntimeError("labenv.apply() after the library was loaded")
        lib.LIB_PATH = p
    handle = lib.load()
    v = os.environ.get("KBNER_GEMM_VARIANT")
    if v:
        handle.kbner_gemm_set_variant(int(v))
    return handle
