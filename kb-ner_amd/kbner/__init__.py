"""kbner: MI355X-native kernels + host engine for KB-NER's XLM-R + CRF token-classification hot path.
The arithmetic lives in libkbner_hip.so (C ABI, include/kbner.h); this package is the thin host side."""
from . import lib  # noqa: F401

__all__ = ["lib"]
