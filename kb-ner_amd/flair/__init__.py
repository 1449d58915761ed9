"""Drop-in `flair` surface for KB-NER's token-classification path, backed by the MI355X kernels in `kbner`.

Only what train.py and the YAML configs exercise on the XLM-R + CRF path is provided: the same
import paths, class names, constructor keywords and method names as the reference's flair/ package, so the
reference's train.py and config/*.yaml run against this package unchanged.  Everything numeric is delegated to
libkbner_hip.so; this package is host-side bookkeeping (reference: flair/__init__.py:7-44)."""
import logging
import os
import sys

import torch

# flair.device semantics of the reference (flair/__init__.py:7-12): first GPU if present, else CPU
if torch.cuda.is_available():
    device = torch.device("cuda:%d" % int(os.environ.get("LOCAL_RANK", "0")))
else:
    device = torch.device("cpu")

cache_root = os.path.expanduser(os.path.join("~", ".flair"))

logger = logging.getLogger("flair")
if not logger.handlers:
    _h = logging.StreamHandler(sys.stdout)
    _h.setFormatter(logging.Formatter("%(asctime)-15s %(message)s"))
    logger.addHandler(_h)
    logger.setLevel(logging.INFO)
    logger.propagate = False

# Launched by torchrun (WORLD_SIZE > 1): one process per GPU over RCCL.  The reference's train.py knows nothing about
# torch.distributed, so the process group is created here, at `import flair`, before any model exists.
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    from kbner import dp as _dp
    _dp.init_from_env()

from . import data  # noqa: E402,F401
from . import models  # noqa: E402,F401
from . import trainers  # noqa: E402,F401
from . import nn  # noqa: E402,F401

__version__ = "0.4.3+kbner.mi355x"
