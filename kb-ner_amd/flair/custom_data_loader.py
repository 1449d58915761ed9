"""Batching for the tagger path.  Behavioural reference: flair/custom_data_loader.py -- BatchedData (:13-19),
ColumnDataLoader.chunk_batches (:84-149: sort ascending by length, cut into batches of `batch_size` sentences when
`sentence_level_batch`, else by a token budget), reshuffle = shuffle BATCH ORDER only (:74-77), assign_tags (:199-378:
per-sentence padded tag tensors + a batch tensor)."""
import random
from typing import List

import torch

from .data import Sentence


class BatchedData(list):
    """a list of sentences plus per-batch feature slots"""

    def __init__(self, sentences):
        super().__init__(sentences)
        self.features = {}
        self.teacher_features = {}
        self.sentence_features = {}
        self.img_features = {}


class ColumnDataLoader:
    def __init__(self, data, batch_size, shuffle=False, args=None, grouped_data=False, use_bert=False, tokenizer=None,
                 sort_data=True, sentence_level_batch=False, model=None):
        self.batch_size = batch_size
        self.args = args
        self.shuffled = shuffle
        self.grouped_data = grouped_data
        self.sentence_level_batch = sentence_level_batch
        self.model = model
        self.use_bert = use_bert
        self.tokenizer = tokenizer
        if use_bert and tokenizer is None:
            raise ValueError("use_bert=True needs a tokenizer (no network to fetch a default one)")
        if sentence_level_batch and batch_size > 500:
            raise AssertionError("batch size too large for sentence-level batching -- wrong batch mode?")
        data = list(data)
        self.num_examples = len(data)
        self.data = self.chunk_batches(data, sort_data=sort_data)

    # -- container
    def __len__(self):
        return len(self.data)

    def __getitem__(self, key):
        if not isinstance(key, int):
            raise TypeError
        if key < 0 or key >= len(self.data):
            raise IndexError
        return self.data[key]

    def __iter__(self):
        for i in range(len(self.data)):
            yield self.data[i]

    def reshuffle(self):
        random.shuffle(self.data)

    def true_reshuffle(self):
        flat = [s for b in self.data for s in b]
        self.data = self.chunk_batches(flat)
        random.shuffle(self.data)
        if getattr(self, "_tag_args", None) is not None:
            self.assign_tags(*self._tag_args)

    def _length(self, item) -> int:
        s = item[0] if self.grouped_data else item
        if self.use_bert:
            return len(self.tokenizer.tokenize(s.to_tokenized_string()))
        return len(s)

    def chunk_batches(self, data, sort_data=True) -> List[list]:
        if sort_data:
            data = sorted(data, key=self._length)
        out, cur, cur_len = [], [], 0
        for x in data:
            n = self._length(x)
            full = len(cur) >= self.batch_size if (self.sentence_level_batch and not self.grouped_data) else (n + cur_len > self.batch_size)
            if full and cur:
                out.append(cur)
                cur, cur_len = [], 0
            cur.append(x)
            cur_len += n
        if cur:
            out.append(cur)
        return out

    def assign_tags(self, tag_type, tag_dictionary, teacher_input=None, grouped_data=False):
        """tag ids as padded int64 tensors: `sentence.<tag_type>_tags` [n_max] per sentence and `batch.<tag_type>_tags`
        [B, n_max] per batch (what _calculate_loss stacks, sequence_tagger_model.py:2434)."""
        self._tag_args = (tag_type, tag_dictionary)
        for i, batch in enumerate(self.data):
            n_max = max(len(s) for s in batch)
            rows = []
            for s in batch:
                ids = [tag_dictionary.get_idx_for_item(t.get_tag(tag_type).value) for t in s]
                row = torch.zeros(n_max, dtype=torch.long)
                row[: len(ids)] = torch.tensor(ids, dtype=torch.long)
                setattr(s, tag_type + "_tags", row)
                rows.append(row)
            b = batch if isinstance(batch, BatchedData) else BatchedData(batch)
            setattr(b, tag_type + "_tags", torch.stack(rows, 0))
            self.data[i] = b

    def assign_embeddings(self):
        """no lookup-table embeddings on this path (Word/Char embeddings are out of scope)"""
        for i, batch in enumerate(self.data):
            if not isinstance(batch, BatchedData):
                self.data[i] = BatchedData(batch)
