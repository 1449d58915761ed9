"""ListCorpus: several datasets trained as one (reference: flair/list_data.py:3-18)."""
from typing import List

from torch.utils.data.dataset import ConcatDataset

from .data import Corpus, FlairDataset


class ListCorpus(Corpus):
    def __init__(self, train: List[FlairDataset], dev: List[FlairDataset], test: List[FlairDataset], name: str = "listcorpus",
                 targets: list = None):
        self.train_list, self.dev_list, self.test_list = train, dev, test
        self._train = ConcatDataset(list(train))
        self._dev = ConcatDataset(list(dev))
        self._test = ConcatDataset(list(test))
        self.name = name
        self.targets = targets if targets is not None else []
