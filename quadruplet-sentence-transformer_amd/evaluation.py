"""Minimal sentence_transformers.evaluation surface used by the reference's evaluators
(models/evaluators.py:9-12,187-216,602-612): the base class, SimilarityFunction, SequentialEvaluator and an
encode()-driven TripletEvaluator. IR metrics / CSV plumbing stay out of scope."""
from __future__ import annotations

import csv
import os
from enum import Enum
from typing import Iterable, List

import numpy as np


class SentenceEvaluator:
    def __call__(self, model, output_path: str = None, epoch: int = -1, steps: int = -1) -> float:
        pass


class SimilarityFunction(Enum):
    COSINE = 0
    EUCLIDEAN = 1
    MANHATTAN = 2
    DOT_PRODUCT = 3


class SequentialEvaluator(SentenceEvaluator):
    """Runs evaluators in order; the main score is main_score_function(scores) (default: the last one)."""

    def __init__(self, evaluators: Iterable[SentenceEvaluator], main_score_function=lambda scores: scores[-1]):
        self.evaluators = list(evaluators)
        self.main_score_function = main_score_function

    def __call__(self, model, output_path: str = None, epoch: int = -1, steps: int = -1) -> float:
        scores = [ev(model, output_path, epoch, steps) for ev in self.evaluators]
        return self.main_score_function(scores)


class TripletEvaluator(SentenceEvaluator):
    """accuracy of d(anchor, positive) < d(anchor, negative) under cosine / manhattan / euclidean distance."""

    def __init__(self, anchors: List[str], positives: List[str], negatives: List[str], main_distance_function=None,
                 name: str = "", batch_size: int = 16, show_progress_bar: bool = False, write_csv: bool = True):
        assert len(anchors) == len(positives) == len(negatives)
        self.anchors, self.positives, self.negatives = anchors, positives, negatives
        self.main_distance_function = main_distance_function
        self.name, self.batch_size, self.show_progress_bar, self.write_csv = name, batch_size, show_progress_bar, write_csv
        self.csv_file = "triplet_evaluation" + ("_" + name if name else "") + "_results.csv"
        self.csv_headers = ["epoch", "steps", "accuracy_cosinus", "accuracy_manhattan", "accuracy_euclidean"]

    def __call__(self, model, output_path: str = None, epoch: int = -1, steps: int = -1) -> float:
        enc = lambda xs: np.asarray(model.encode(xs, batch_size=self.batch_size, show_progress_bar=self.show_progress_bar,
                                                 convert_to_numpy=True), dtype=np.float64)
        a, p, n = enc(self.anchors), enc(self.positives), enc(self.negatives)

        def cosd(x, y):
            return 1.0 - (x * y).sum(1) / (np.linalg.norm(x, axis=1) * np.linalg.norm(y, axis=1) + 1e-30)

        acc_cos = float(np.mean(cosd(a, p) < cosd(a, n)))
        acc_man = float(np.mean(np.abs(a - p).sum(1) < np.abs(a - n).sum(1)))
        acc_euc = float(np.mean(np.linalg.norm(a - p, axis=1) < np.linalg.norm(a - n, axis=1)))
        if output_path is not None and self.write_csv:
            path = os.path.join(output_path, self.csv_file)
            new = not os.path.isfile(path)
            with open(path, "a", newline="", encoding="utf-8") as f:
                w = csv.writer(f)
                if new:
                    w.writerow(self.csv_headers)
                w.writerow([epoch, steps, acc_cos, acc_man, acc_euc])
        if self.main_distance_function == SimilarityFunction.COSINE:
            return acc_cos
        if self.main_distance_function == SimilarityFunction.MANHATTAN:
            return acc_man
        if self.main_distance_function == SimilarityFunction.EUCLIDEAN:
            return acc_euc
        return max(acc_cos, acc_man, acc_euc)
