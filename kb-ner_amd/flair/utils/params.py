"""YAML -> Params (reference: flair/utils/params.py:98-109: yaml.load of the file, dict-like access)."""
import copy

import yaml


class Params(dict):
    """dict with the couple of accessors the config parser uses"""

    def __init__(self, params=None):
        super().__init__(params or {})
        self.params = self

    def as_dict(self):
        return copy.deepcopy(dict(self))

    def pop_key(self, key, default=None):
        return self.pop(key, default)

    @staticmethod
    def from_file(params_file: str, params_overrides: str = "") -> "Params":
        with open(params_file, encoding="utf-8") as f:
            d = yaml.load(f, Loader=yaml.FullLoader) or {}
        if params_overrides:
            over = yaml.load(params_overrides, Loader=yaml.FullLoader) or {}
            d = dict_merge(d, over)
        return Params(d)


def dict_merge(a: dict, b: dict) -> dict:
    """recursive merge, b wins (reference: flair/algorithms/dict_merge.py)"""
    out = dict(a)
    for k, v in b.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = dict_merge(out[k], v)
        else:
            out[k] = v
    return out
