"""CoNLL column readers.  Behavioural reference: flair/datasets.py ColumnCorpus (:21-136) and ColumnDataset (:852-1004):
whitespace-split columns, lines starting with `comment_symbol` skipped, blank line = sentence end, optional
IOB -> IOBES conversion of one tag column (every `B-X` context token of a KB-NER file becomes `S-X`)."""
import logging
import re
from pathlib import Path
from typing import Dict, List, Union

import torch.utils.data as _torch_data

from .data import Corpus, FlairDataset, Sentence, Token

log = logging.getLogger("flair")


class ColumnDataset(FlairDataset):
    def __init__(self, path_to_column_file: Union[str, Path], column_name_map: Dict[int, str], tag_to_bioes: str = None,
                 comment_symbol: str = None, in_memory: bool = True):
        path = Path(path_to_column_file)
        assert path.exists(), "%s does not exist" % path
        self.path_to_column_file = path
        self.column_name_map = column_name_map
        self.tag_to_bioes = tag_to_bioes
        self.comment_symbol = comment_symbol
        self.in_memory = True  # always materialised (the corpora on this path are small)
        self.text_column = next((c for c, n in column_name_map.items() if n == "text"), 0)
        self.sentences: List[Sentence] = []
        try:
            text = path.read_text(encoding="utf-8")
        except UnicodeDecodeError:
            log.info('UTF-8 can\'t read: %s ... using "latin-1" instead.', path)
            text = path.read_text(encoding="latin1")
        cur = Sentence()
        for line in text.split("\n"):
            if comment_symbol is not None and line.startswith(comment_symbol):
                continue
            if line.strip() == "":
                self._close(cur)
                cur = Sentence()
                continue
            fields = re.split(r"\s+", line.strip("\r"))
            tok = Token(fields[self.text_column])
            for col, name in column_name_map.items():
                if col != self.text_column and len(fields) > col:
                    tok.add_tag(name, fields[col])
            cur.add_token(tok)
        self._close(cur)
        self.total_sentence_count = len(self.sentences)

    def _close(self, sentence: Sentence):
        if len(sentence) == 0:
            return
        sentence.infer_space_after()
        if self.tag_to_bioes is not None:
            sentence.convert_tag_scheme(tag_type=self.tag_to_bioes, target_scheme="iobes")
        self.sentences.append(sentence)

    @property
    def reset_sentence_count(self):
        self.total_sentence_count = len(self.sentences)

    def is_in_memory(self) -> bool:
        return True

    def __len__(self):
        return self.total_sentence_count

    def __getitem__(self, index: int = 0) -> Sentence:
        return self.sentences[index]

    def __iter__(self):
        return iter(self.sentences[: self.total_sentence_count])


class ColumnCorpus(Corpus):
    def __init__(self, data_folder: Union[str, Path], column_format: Dict[int, str], train_file=None, test_file=None,
                 dev_file=None, tag_to_bioes=None, comment_symbol: str = None, in_memory: bool = True):
        folder = Path(data_folder)
        train_file = folder / train_file if train_file is not None else None
        test_file = folder / test_file if test_file is not None else None
        dev_file = folder / dev_file if dev_file is not None else None
        if train_file is None:  # discover by name, like the reference (:57-80)
            skip = (".gz", ".swp", ".pkl")
            for f in sorted(folder.iterdir()):
                n = f.name
                if n.endswith(skip) or not f.is_file():
                    continue
                if "train" in n:
                    train_file = f
                if "dev" in n or "testa" in n:
                    dev_file = f
                if "testb" in n:
                    test_file = f
            if test_file is None:
                for f in sorted(folder.iterdir()):
                    if "test" in f.name and not f.name.endswith(".gz") and f.is_file():
                        test_file = f
        log.info("Reading data from %s", folder)
        log.info("Train: %s", train_file)
        log.info("Dev: %s", dev_file)
        log.info("Test: %s", test_file)
        mk = lambda p: ColumnDataset(p, column_format, tag_to_bioes, comment_symbol=comment_symbol, in_memory=in_memory)  # noqa: E731
        train = mk(train_file)
        # no test / dev file: carve 10% of train off the end (deterministic here; the reference samples at random)
        if test_file is not None:
            test = mk(test_file)
        else:
            test, train = _split_tail(train)
        if dev_file is not None:
            dev = mk(dev_file)
        else:
            dev, train = _split_tail(train)
        super().__init__(train, dev, test, name=str(folder))


class _Subset(FlairDataset):
    def __init__(self, sentences):
        self.sentences = sentences
        self.total_sentence_count = len(sentences)

    def __len__(self):
        return len(self.sentences)

    def __getitem__(self, i):
        return self.sentences[i]

    def __iter__(self):
        return iter(self.sentences)


def _split_tail(ds, frac=0.1):
    n = len(ds)
    k = max(1, round(n * frac)) if n > 1 else 0
    sents = [ds[i] for i in range(n)]
    return _Subset(sents[n - k:]), _Subset(sents[: n - k])


class DataLoader(_torch_data.DataLoader):
    """Sentence-list loader (`from flair.datasets import DataLoader`, train.py:25; reference: flair/datasets.py:4729-4771):
    batches are plain python lists of Sentence; in-memory datasets are never handed to worker processes."""

    def __init__(self, dataset, batch_size=1, shuffle=False, sampler=None, batch_sampler=None, num_workers=4, drop_last=False,
                 timeout=0, worker_init_fn=None):
        inner = dataset
        while isinstance(inner, (_torch_data.Subset, _torch_data.ConcatDataset)):
            inner = inner.dataset if isinstance(inner, _torch_data.Subset) else inner.datasets[0]
        if isinstance(inner, list) or (isinstance(inner, FlairDataset) and inner.is_in_memory()):
            num_workers = 0
        if batch_sampler is not None:  # torch forbids batch_size / shuffle / drop_last next to a batch_sampler
            super().__init__(dataset, batch_sampler=batch_sampler, num_workers=num_workers, collate_fn=list, timeout=timeout,
                             worker_init_fn=worker_init_fn)
        else:
            super().__init__(dataset, batch_size=batch_size, shuffle=shuffle, sampler=sampler, num_workers=num_workers,
                             collate_fn=list, drop_last=drop_last, timeout=timeout, worker_init_fn=worker_init_fn)
