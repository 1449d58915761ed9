"""TEST INFRASTRUCTURE (container only) -- import the read-only reference at /root/reference.

This module is used ONLY by oracle/gen_golden.py to produce the committed fixtures under
tests/golden/.  It never travels to the GPU box as a dependency of anything: /root/reference
does not exist there, and nothing under tests -m gpu / bench.py / smoke() imports this file.

The reference pins torch==1.3.1 / transformers==3.0.0 (requirements.txt:28,30); this container
has torch 2.10 / transformers 5.15, so a few shims are applied *before* import.
Nothing of the reference is copied or modified.
"""
import importlib.abc
import importlib.machinery
import sys
import types

REFERENCE_ROOT = "/root/reference"

_MISSING = (
    "segtok", "gensim", "bpemb", "deprecated", "pytorch_transformers", "h5py", "allennlp",
    "hyperopt", "mpld3", "overrides", "pyhocon", "boto3", "botocore", "nltk", "conllu",
    "torch_struct", "tensorboardX", "apex", "spacy", "tabulate_stub", "matplotlib", "sklearn_stub",
    "langdetect", "ftfy", "sqlitedict", "lmdb", "elasticsearch", "konoha", "tiny_tokenizer",
    "mecab", "janome", "bs4",
)


class _StubClass:
    """Every attribute of a stub module is a class, so `class X(stub.Y)` works."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, name):
        raise AttributeError(name)


class _StubModule(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        if name == "deprecated":  # `from deprecated import deprecated` must be a pass-through decorator
            def deprecated(*dargs, **dkw):
                if len(dargs) == 1 and callable(dargs[0]) and not dkw:
                    return dargs[0]
                return lambda f: f
            return deprecated
        cls = type(name, (_StubClass,), {})
        setattr(self, name, cls)
        return cls


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        top = fullname.split(".")[0]
        if top in _MISSING:
            try:
                # only stub what really is missing
                for f in sys.meta_path:
                    if f is self:
                        continue
                    spec = f.find_spec(fullname, path, target) if hasattr(f, "find_spec") else None
                    if spec is not None:
                        return None
            except Exception:
                pass
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        pass


class HFAdamW:
    """Restatement of transformers==3.0.0 `AdamW` semantics for the shim (optimization.py in that
    release): eps default 1e-6, weight_decay 0, correct_bias True; decoupled decay applied AFTER the
    Adam update with the group's lr.  (torch.optim.AdamW differs in eps/decay defaults and order.)"""

    def __new__(cls, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        import torch

        class _Impl(torch.optim.Optimizer):
            def __init__(self, params, lr, betas, eps, weight_decay, correct_bias):
                super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                              correct_bias=correct_bias))

            @torch.no_grad()
            def step(self, closure=None):
                import math
                for group in self.param_groups:
                    for p in group["params"]:
                        if p.grad is None:
                            continue
                        g = p.grad
                        st = self.state[p]
                        if len(st) == 0:
                            st["step"] = 0
                            st["exp_avg"] = torch.zeros_like(p)
                            st["exp_avg_sq"] = torch.zeros_like(p)
                        m, v = st["exp_avg"], st["exp_avg_sq"]
                        b1, b2 = group["betas"]
                        st["step"] += 1
                        m.mul_(b1).add_(g, alpha=1.0 - b1)
                        v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
                        denom = v.sqrt().add_(group["eps"])
                        step_size = group["lr"]
                        if group["correct_bias"]:
                            bc1 = 1.0 - b1 ** st["step"]
                            bc2 = 1.0 - b2 ** st["step"]
                            step_size = step_size * math.sqrt(bc2) / bc1
                        p.addcdiv_(m, denom, value=-step_size)
                        if group["weight_decay"] > 0.0:
                            p.add_(p, alpha=-group["lr"] * group["weight_decay"])

        return _Impl(params, lr, betas, eps, weight_decay, correct_bias)


class TokenizerAdapter:
    """transformers-3.0.0 tokenizer surface the reference uses (flair/embeddings.py:3001-3004,3143-3163,3173,3200-3227) over
    a transformers-5.x fast tokenizer, which has neither `_eos_token` nor `encode_plus(ids, ...)` (without
    them the reference silently takes the `else` branch at :3228-3230 and feeds ids WITHOUT <s> </s> while still using
    begin_offset=1).  encode_plus restates the 3.0.0 semantics for a list of ids: `<s> ids[:max_length-2] </s>`; with
    return_overflowing_tokens the overflow is the tail that did not fit, preceded by `stride` ids of overlap."""

    # "3.0.0": the pinned release's `longest_first` loop (see encode_plus); "intended": what that loop was meant to do and what
    # every later release does for a single sequence -- the overflow is the tail starting `stride` ids before the cut.
    # oracle/gen_golden_windows.py captures the reference's sliding-window path under BOTH.
    OVERFLOW = "3.0.0"

    def __init__(self, tok):
        self.__dict__["_tok"] = tok
        for a in ("eos", "sep", "bos", "cls", "pad", "unk"):
            self.__dict__["_%s_token" % a] = getattr(tok, a + "_token", None)

    def __getattr__(self, name):
        return getattr(self.__dict__["_tok"], name)

    def __len__(self):
        return len(self._tok)

    def tokenize(self, text, **kw):
        return self._tok.tokenize(text, **kw)

    def convert_tokens_to_ids(self, toks):
        return self._tok.convert_tokens_to_ids(toks)

    def encode_plus(self, ids, max_length=None, stride=0, return_overflowing_tokens=False, truncation=True, **kw):
        ids = list(ids)
        bos, eos = self._tok.bos_token_id, self._tok.eos_token_id
        if bos is None:
            bos = self._tok.cls_token_id
        if eos is None:
            eos = self._tok.sep_token_id
        out = {}
        room = len(ids) if max_length is None else max_length - 2
        if truncation and len(ids) > room:
            # truncation=True selects 'longest_first', whose 3.0.0 loop removes ONE id at a time from the end: the first
            # removal records the last stride+1 ids, every later one appends the id it removes -- so the overflow is the
            # window's tail followed by the earlier ids in REVERSE order (a defect of that release, fixed later; restated from
            # the published 3.0.0 algorithm, which cannot be installed here).  The build's product path implements the intended
            # semantics (next window restarts `stride` ids before the cut); KB-NER data never exceeds one window in training
            # (kb/context_process.py:974), so goldens captured through this adapter never reach this branch.
            if TokenizerAdapter.OVERFLOW == "intended":
                overflow = ids[max(room - stride, 0):]
                ids = ids[:room]
            else:
                overflow = []
                for _ in range(len(ids) - room):
                    w = min(len(ids), stride + 1) if not overflow else 1
                    overflow.extend(ids[-w:])
                    ids = ids[:-1]
            if return_overflowing_tokens:
                out["overflowing_tokens"] = overflow
        out["input_ids"] = [bos] + ids + [eos]
        return out

    def save_pretrained(self, path, **kw):
        return self._tok.save_pretrained(path, **kw)


def wrap_auto_tokenizer():
    """AutoTokenizer.from_pretrained (flair/embeddings.py:2951) returns the adapter"""
    import transformers
    if getattr(transformers.AutoTokenizer, "_kbner_wrapped", False):
        return
    orig = transformers.AutoTokenizer.from_pretrained

    def from_pretrained(*a, **k):
        return TokenizerAdapter(orig(*a, **k))

    transformers.AutoTokenizer.from_pretrained = staticmethod(from_pretrained)
    transformers.AutoTokenizer._kbner_wrapped = True


_installed = False


def install():
    """Apply the shims and put /root/reference on sys.path.  Idempotent."""
    global _installed
    if _installed:
        return
    _installed = True
    import torch
    import yaml
    import transformers

    sys.meta_path.append(_StubFinder())
    # transformers 5.x removed AdamW; set on the lazy-module CLASS
    try:
        type(transformers).AdamW = HFAdamW
    except Exception:
        transformers.AdamW = HFAdamW
    # hard-coded .cuda() (sequence_tagger_model.py:1028,2555-2556)
    torch.Tensor.cuda = lambda self, *a, **k: self
    # yaml.load without Loader (flair/utils/params.py:104)
    _orig_yaml_load = yaml.load

    def _yaml_load(stream, Loader=None, **kw):
        return _orig_yaml_load(stream, Loader=Loader or yaml.FullLoader, **kw)

    yaml.load = _yaml_load
    # torch.load of pickled model objects (flair/nn.py:101)
    _orig_torch_load = torch.load

    def _torch_load(*a, **k):
        k.setdefault("weights_only", False)
        return _orig_torch_load(*a, **k)

    torch.load = _torch_load
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def load_reference():
    """Returns the reference `flair` package (imported from /root/reference)."""
    install()
    # make sure we import the REFERENCE's flair and not the build's drop-in of the same name
    for name in list(sys.modules):
        if name == "flair" or name.startswith("flair."):
            mod = sys.modules[name]
            if not getattr(mod, "__file__", "") or not str(mod.__file__).startswith(REFERENCE_ROOT):
                del sys.modules[name]
    import flair  # noqa
    assert flair.__file__.startswith(REFERENCE_ROOT), flair.__file__
    return flair
