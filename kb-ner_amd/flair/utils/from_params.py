from .params import Params  # noqa: F401  (train.py imports Params from flair.utils.from_params)
