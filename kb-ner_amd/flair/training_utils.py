"""Metric / Result / logging helpers.  Behavioural reference: flair/training_utils.py -- Result (:15-24), Metric
(:26-187: per-class tp/fp/fn/tn, every ratio rounded to 4 decimals BEFORE it is combined, macro F = unrounded mean
of the rounded per-class F), EvaluationMetric (:270-276), store_embeddings (:346-368), add_file_handler (:336-343)."""
import itertools
import logging
from collections import defaultdict
from enum import Enum
from pathlib import Path
from typing import List


class Result(object):
    def __init__(self, main_score: float, log_header: str, log_line: str, detailed_results: str, macro_score: float = None):
        self.main_score = main_score
        self.log_header = log_header
        self.log_line = log_line
        self.detailed_results = detailed_results
        self.macro_score = macro_score


def _ratio(num, den):
    return round(num / den, 4) if den > 0 else 0.0


class Metric(object):
    def __init__(self, name):
        self.name = name
        self._tps, self._fps, self._tns, self._fns = (defaultdict(int) for _ in range(4))

    def add_tp(self, class_name):
        self._tps[class_name] += 1

    def add_tn(self, class_name):
        self._tns[class_name] += 1

    def add_fp(self, class_name):
        self._fps[class_name] += 1

    def add_fn(self, class_name):
        self._fns[class_name] += 1

    def _get(self, table, class_name):
        if class_name is None:
            return sum(table[c] for c in self.get_classes())
        return table[class_name]

    def get_tp(self, class_name=None):
        return self._get(self._tps, class_name)

    def get_tn(self, class_name=None):
        return self._get(self._tns, class_name)

    def get_fp(self, class_name=None):
        return self._get(self._fps, class_name)

    def get_fn(self, class_name=None):
        return self._get(self._fns, class_name)

    def precision(self, class_name=None):
        tp = self.get_tp(class_name)
        return _ratio(tp, tp + self.get_fp(class_name))

    def recall(self, class_name=None):
        tp = self.get_tp(class_name)
        return _ratio(tp, tp + self.get_fn(class_name))

    def f_score(self, class_name=None):
        p, r = self.precision(class_name), self.recall(class_name)
        return round(2 * (p * r) / (p + r), 4) if p + r > 0 else 0.0

    def accuracy(self, class_name=None):
        tp = self.get_tp(class_name)
        return _ratio(tp, tp + self.get_fp(class_name) + self.get_fn(class_name))

    def micro_avg_f_score(self):
        return self.f_score(None)

    def macro_avg_f_score(self):
        fs = [self.f_score(c) for c in self.get_classes()]
        return sum(fs) / len(fs) if fs else 0.0

    def micro_avg_accuracy(self):
        return self.accuracy(None)

    def macro_avg_accuracy(self):
        acc = [self.accuracy(c) for c in self.get_classes()]
        return round(sum(acc) / len(acc), 4) if acc else 0.0

    # ---- data-parallel evaluation: every rank scores its share of the batches, the per-class counters are
    # summed over ranks, and precision / recall / F1 are computed from the totals -- exactly what one rank scoring every batch
    # gets, because every score of this class is a function of the four counter tables only
    def counts(self):
        """{class: [tp, fp, tn, fn]} (plain ints: picklable for all_gather_object)"""
        return {c: [self._tps[c], self._fps[c], self._tns[c], self._fns[c]] for c in self.get_classes()}

    def merge_counts(self, counts):
        for c, (tp, fp, tn, fn) in counts.items():
            self._tps[c] += tp
            self._fps[c] += fp
            self._tns[c] += tn
            self._fns[c] += fn
        return self

    def get_classes(self) -> List:
        keys = set(itertools.chain(self._tps, self._fps, self._tns, self._fns))
        return sorted(k for k in keys if k is not None)

    def to_tsv(self):
        return "{}\t{}\t{}\t{}".format(self.precision(), self.recall(), self.accuracy(), self.micro_avg_f_score())

    @staticmethod
    def tsv_header(prefix=None):
        if prefix:
            return "{0}_PRECISION\t{0}_RECALL\t{0}_ACCURACY\t{0}_F-SCORE".format(prefix)
        return "PRECISION\tRECALL\tACCURACY\tF-SCORE"

    @staticmethod
    def to_empty_tsv():
        return "\t_\t_\t_\t_"

    def __str__(self):
        lines = []
        for c in [None] + self.get_classes():
            lines.append("{0:<10}\ttp: {1} - fp: {2} - fn: {3} - tn: {4} - precision: {5:.4f} - recall: {6:.4f} - "
                         "accuracy: {7:.4f} - f1-score: {8:.4f}".format(self.name if c is None else c, self.get_tp(c),
                                                                         self.get_fp(c), self.get_fn(c), self.get_tn(c),
                                                                         self.precision(c), self.recall(c), self.accuracy(c),
                                                                         self.f_score(c)))
        return "\n".join(lines)


class EvaluationMetric(Enum):
    MICRO_ACCURACY = "micro-average accuracy"
    MICRO_F1_SCORE = "micro-average f1-score"
    MACRO_ACCURACY = "macro-average accuracy"
    MACRO_F1_SCORE = "macro-average f1-score"
    MEAN_SQUARED_ERROR = "mean squared error"


def init_output_file(base_path: Path, file_name: str) -> Path:
    base_path = Path(base_path)
    base_path.mkdir(parents=True, exist_ok=True)
    f = base_path / file_name
    open(f, "w", encoding="utf-8").close()
    return f


def add_file_handler(log, output_file):
    Path(output_file).parent.mkdir(parents=True, exist_ok=True)
    fh = logging.FileHandler(output_file, mode="w", encoding="utf-8")
    fh.setLevel(logging.INFO)
    fh.setFormatter(logging.Formatter("%(asctime)-15s %(message)s"))
    log.addHandler(fh)
    return fh


def log_line(log):
    log.info("-" * 100)


def store_embeddings(sentences, storage_mode: str):
    """fine-tuned transformer embeddings are never cached (they change every step): drop per-token vectors and
    per-batch features, which is what 'none' / 'cpu' amount to on this path (training_utils.py:346-368)."""
    for s in sentences:
        if hasattr(s, "clear_embeddings"):
            s.clear_embeddings()
    if hasattr(sentences, "features"):
        sentences.features = {}
