"""Character language model container for FlairEmbeddings (inference only).

Behavioural reference (restated): flair/models/language_model.py -- the saved-model format of LanguageModel.save (:230-243: a
torch pickle with state_dict, dictionary, is_forward_lm, hidden_size, nlayers, embedding_size, nout, dropout) and
load_language_model (:161-178).  The arithmetic (embedding -> 1-layer LSTM -> hidden states, :71-138) runs in
kbner.stack.CharLM on the HIP LSTM kernel; this class only carries the weights and the character dictionary.  Training
character LMs is out of scope (language_model trainer)."""
from pathlib import Path
from typing import Union

import torch


class LanguageModel:
    def __init__(self, dictionary, is_forward_lm: bool, hidden_size: int, nlayers: int, embedding_size: int = 100, nout=None,
                 dropout=0.1, state_dict=None):
        if nlayers != 1:
            raise NotImplementedError("character LMs with more than one LSTM layer are not supported (the shipped Flair LMs have 1)")
        if nout is not None:
            raise NotImplementedError("character LMs with an output projection (nout) are not supported")
        self.dictionary = dictionary
        self.is_forward_lm = bool(is_forward_lm)
        self.hidden_size, self.nlayers, self.embedding_size, self.nout, self.dropout = hidden_size, nlayers, embedding_size, nout, dropout
        self._state = {k: v.detach().float().cpu() for k, v in (state_dict or {}).items()}

    def state_dict(self):
        return self._state

    @classmethod
    def load_language_model(cls, model_file: Union[Path, str]):
        state = torch.load(str(model_file), map_location="cpu", weights_only=False)
        return cls(state["dictionary"], state["is_forward_lm"], state["hidden_size"], state["nlayers"], state["embedding_size"],
                   state["nout"], state["dropout"], state_dict=state["state_dict"])

    def save(self, file):
        torch.save({"state_dict": self._state, "dictionary": self.dictionary, "is_forward_lm": self.is_forward_lm,
                    "hidden_size": self.hidden_size, "nlayers": self.nlayers, "embedding_size": self.embedding_size,
                    "nout": self.nout, "dropout": self.dropout}, str(file), pickle_protocol=4)

    def eval(self):
        return self

    def to(self, *a, **k):
        return self
