"""Model base: save / load of the tagger state (reference: flair/nn.py:15-139).  The reference pickles the embeddings
OBJECT into best-model.pt; here the state dict carries plain tensors + constructor arguments, which is what this
package's `load` needs and keeps the file loadable without importing reference classes."""
import warnings
from pathlib import Path
from typing import Union

import torch

import flair


class Model(torch.nn.Module):
    def _get_state_dict(self):
        raise NotImplementedError

    @classmethod
    def _init_model_with_state_dict(cls, state):
        raise NotImplementedError

    def save(self, model_file: Union[str, Path]):
        torch.save(self._get_state_dict(), str(model_file), pickle_protocol=4)

    def save_checkpoint(self, model_file, optimizer_state: dict, scheduler_state: dict, epoch: int, loss: float):
        st = self._get_state_dict()
        st.update(optimizer_state_dict=optimizer_state, scheduler_state_dict=scheduler_state, epoch=epoch, loss=loss)
        torch.save(st, str(model_file), pickle_protocol=4)

    @classmethod
    def load(cls, model_file: Union[str, Path], device: str = None):
        with warnings.catch_warnings():
            warnings.filterwarnings("ignore")
            state = torch.load(str(model_file), map_location="cpu", weights_only=False)
        model = cls._init_model_with_state_dict(state)
        model.eval()
        return model

    @classmethod
    def load_checkpoint(cls, checkpoint_file, device=None):
        state = torch.load(str(checkpoint_file), map_location="cpu", weights_only=False)
        model = cls._init_model_with_state_dict(state)
        return {"model": model, "epoch": state.get("epoch"), "loss": state.get("loss"),
                "optimizer_state_dict": state.get("optimizer_state_dict"), "scheduler_state_dict": state.get("scheduler_state_dict")}
