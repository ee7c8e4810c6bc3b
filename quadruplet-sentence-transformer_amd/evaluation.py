"""sentence_transformers.evaluation surface used by the reference's evaluators (models/evaluators.py:9-12,187-216,
572-612; ir_evauation_script.py:107-131): the base class, SimilarityFunction, SequentialEvaluator, an encode()-driven
TripletEvaluator and InformationRetrievalEvaluator, whose scoring + top-k run on the GPU
through libqst (util.topk_scores)."""
from __future__ import annotations

import csv
import os
from enum import Enum
from typing import Callable, Dict, Iterable, List, Optional, Set

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


def ir_metrics(queries_result_list: List[List[dict]], queries_ids: List[str], relevant_docs: Dict[str, Set[str]],
               mrr_at_k: List[int], ndcg_at_k: List[int], accuracy_at_k: List[int], precision_recall_at_k: List[int],
               map_at_k: List[int]) -> dict:
    """Accuracy@k, Precision@k, Recall@k, MRR@k, NDCG@k (binary gains, log2 discount), MAP@k over ranked hit lists
    [{'corpus_id', 'score'}] (best first), with the definitions InformationRetrievalEvaluator.compute_metrics uses:
    MAP@k divides by min(k, |relevant|); a query with no hit in the top k contributes 0."""
    n = len(queries_ids)
    acc = {k: 0 for k in accuracy_at_k}
    prec = {k: [] for k in precision_recall_at_k}
    rec = {k: [] for k in precision_recall_at_k}
    mrr = {k: 0.0 for k in mrr_at_k}
    ndcg = {k: [] for k in ndcg_at_k}
    ap = {k: [] for k in map_at_k}
    for qi, qid in enumerate(queries_ids):
        hits = sorted(queries_result_list[qi], key=lambda h: h["score"], reverse=True)
        rel = relevant_docs[qid]
        flags = [h["corpus_id"] in rel for h in hits]
        for k in accuracy_at_k:
            acc[k] += int(any(flags[:k]))
        for k in precision_recall_at_k:
            c = sum(flags[:k])
            prec[k].append(c / k)
            rec[k].append(c / len(rel))
        for k in mrr_at_k:
            for rank, f in enumerate(flags[:k]):
                if f:
                    mrr[k] += 1.0 / (rank + 1)
                    break
        for k in ndcg_at_k:
            dcg = sum(1.0 / np.log2(r + 2) for r, f in enumerate(flags[:k]) if f)
            idcg = sum(1.0 / np.log2(r + 2) for r in range(min(k, len(rel))))
            ndcg[k].append(dcg / idcg if idcg > 0 else 0.0)
        for k in map_at_k:
            good, s_prec = 0, 0.0
            for rank, f in enumerate(flags[:k]):
                if f:
                    good += 1
                    s_prec += good / (rank + 1)
            ap[k].append(s_prec / min(k, len(rel)))
    return {"accuracy@k": {k: acc[k] / n for k in acc}, "precision@k": {k: float(np.mean(v)) for k, v in prec.items()},
            "recall@k": {k: float(np.mean(v)) for k, v in rec.items()}, "ndcg@k": {k: float(np.mean(v)) for k, v in ndcg.items()},
            "mrr@k": {k: mrr[k] / n for k in mrr}, "map@k": {k: float(np.mean(v)) for k, v in ap.items()}}


_KNOWN_SCORE_NAMES = {"cos_sim": 1, "dot_score": 0, "euclid_score": 2, "euclidean_score": 2}


def resolve_score_function(name: str, fn: Optional[Callable], device=None):
    """How one `score_functions` entry is evaluated: ("native", mode) = the fused libqst score + top-k kernel,
    ("callable", fn) = call fn(query_emb, corpus_emb) -> [nq, nc] and select with qst_topk_rows.

    Native is chosen by BEHAVIOUR, never by the dictionary key alone: this package's own util.cos_sim / dot_score /
    euclidean_score carry a mode tag; any other callable (e.g. the reference's `euclidean_score`,
    /root/reference/models/evaluators.py:392-405, which arrives as a foreign function object) is run once on a small
    fixed probe and taken over by the native mode whose score matrix it reproduces; a callable that matches none of
    them -- including a custom function registered under the name 'cos_sim' -- is called as given. `None` is accepted
    for the three known names."""
    import torch
    if fn is None:
        if name not in _KNOWN_SCORE_NAMES:
            raise ValueError(f"score function {name!r} is None: give a callable, or one of {sorted(_KNOWN_SCORE_NAMES)}")
        return ("native", _KNOWN_SCORE_NAMES[name])
    if not callable(fn):
        raise ValueError(f"score function {name!r} must be callable (got {type(fn).__name__})")
    mode = getattr(fn, "_qst_mode", None)
    if mode is not None:
        return ("native", int(mode))
    from . import util
    try:
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        g = torch.Generator().manual_seed(20240229)
        q = (torch.randn(6, 32, generator=g) * 1.7 + 0.3).to(dev)
        c = (torch.randn(9, 32, generator=g) * 0.6 - 0.2).to(dev)
        got = torch.as_tensor(fn(q, c)).to(dev, torch.float32)
        if tuple(got.shape) == (6, 9):
            for m in (1, 0, 2):
                ref = util.score_matrix(q, c, m)
                # split-bf16 x3 products are accurate to ~2^-17 of sum |a_k b_k|, i.e. relative to the matrix's scale,
                # not to each (possibly cancelling) entry; the three native modes differ by O(1) on this probe
                if float((got - ref).abs().max()) <= 2e-4 * max(1.0, float(ref.abs().max())):
                    return ("native", m)
    except Exception:          # a callable that cannot take the probe is simply called on the real embeddings later
        pass
    return ("callable", fn)


class InformationRetrievalEvaluator(SentenceEvaluator):
    """Queries against a corpus: encode both, score, keep the top max(k) hits per query across corpus chunks, report
    Accuracy/Precision/Recall/MRR/NDCG/MAP @k. Constructor, CSV layout and return value follow sentence-transformers
    2.2.2 as the reference drives it (models/evaluators.py:572-588, ir_evauation_script.py:107-131): the main score is
    max over score functions of MAP@max(map_at_k) unless main_score_function names one.

    score_functions is any {name: callable} dictionary, as in ST -- the reference always passes
    {'cos_sim': cos_sim, 'dot_score': dot_score, 'euclid_score': euclidean_score} (training/main.py:57,
    ir_evauation_script.py:71). Entries that are, or behave like, cosine / dot / 1/(1+L2) scoring run as ONE fused
    libqst call per corpus chunk (qst_topk_scores); any other callable is called and its matrix goes through
    qst_topk_rows (see resolve_score_function). Per-chunk results are merged by a second top-k over the candidates;
    only the metric arithmetic is host Python."""

    def __init__(self, queries: Dict[str, str], corpus: Dict[str, str], relevant_docs: Dict[str, Set[str]],
                 corpus_chunk_size: int = 50000, mrr_at_k: List[int] = [10], ndcg_at_k: List[int] = [10],
                 accuracy_at_k: List[int] = [1, 3, 5, 10], precision_recall_at_k: List[int] = [1, 3, 5, 10],
                 map_at_k: List[int] = [100], show_progress_bar: bool = False, batch_size: int = 32, name: str = "",
                 write_csv: bool = True, score_functions: Optional[Dict[str, Callable]] = None,
                 main_score_function: Optional[str] = None):
        self.queries_ids = [qid for qid in queries if qid in relevant_docs and len(relevant_docs[qid]) > 0]
        self.queries = [queries[qid] for qid in self.queries_ids]
        self.corpus_ids = list(corpus.keys())
        self.corpus = [corpus[cid] for cid in self.corpus_ids]
        self.relevant_docs = relevant_docs
        self.corpus_chunk_size = corpus_chunk_size
        self.mrr_at_k, self.ndcg_at_k, self.accuracy_at_k = mrr_at_k, ndcg_at_k, accuracy_at_k
        self.precision_recall_at_k, self.map_at_k = precision_recall_at_k, map_at_k
        self.show_progress_bar, self.batch_size, self.name, self.write_csv = show_progress_bar, batch_size, name, write_csv
        if score_functions is None:              # ST's default: {'cos_sim': cos_sim, 'dot_score': dot_score}
            score_functions = {"cos_sim": None, "dot_score": None}
        self.score_functions = dict(score_functions)
        self.score_function_names = sorted(self.score_functions.keys())
        for nm, fn in self.score_functions.items():
            if fn is None and nm not in _KNOWN_SCORE_NAMES:
                raise ValueError(f"score function {nm!r} is None: give a callable, or one of {sorted(_KNOWN_SCORE_NAMES)}")
            if fn is not None and not callable(fn):
                raise ValueError(f"score function {nm!r} must be callable (got {type(fn).__name__})")
        self._resolved = None                    # name -> ("native", mode) | ("callable", fn); needs the device
        self.main_score_function = main_score_function
        self.csv_file = "Information-Retrieval_evaluation" + ("_" + name if name else "") + "_results.csv"
        self.csv_headers = ["epoch", "steps"]
        for nm in self.score_function_names:
            for k in accuracy_at_k:
                self.csv_headers.append(f"{nm}-Accuracy@{k}")
            for k in precision_recall_at_k:
                self.csv_headers.append(f"{nm}-Precision@{k}")
                self.csv_headers.append(f"{nm}-Recall@{k}")
            for k in mrr_at_k:
                self.csv_headers.append(f"{nm}-MRR@{k}")
            for k in ndcg_at_k:
                self.csv_headers.append(f"{nm}-NDCG@{k}")
            for k in map_at_k:
                self.csv_headers.append(f"{nm}-MAP@{k}")

    def __call__(self, model, output_path: str = None, epoch: int = -1, steps: int = -1, *args, **kwargs) -> float:
        scores = self.compute_metrices(model, *args, **kwargs)
        if output_path is not None and self.write_csv:
            path = os.path.join(output_path, self.csv_file)
            new = not os.path.isfile(path)
            with open(path, "a", newline="", encoding="utf-8") as f:
                w = csv.writer(f)
                if new:
                    w.writerow(self.csv_headers)
                row = [epoch, steps]
                for nm in self.score_function_names:
                    for k in self.accuracy_at_k:
                        row.append(scores[nm]["accuracy@k"][k])
                    for k in self.precision_recall_at_k:
                        row.append(scores[nm]["precision@k"][k])
                        row.append(scores[nm]["recall@k"][k])
                    for k in self.mrr_at_k:
                        row.append(scores[nm]["mrr@k"][k])
                    for k in self.ndcg_at_k:
                        row.append(scores[nm]["ndcg@k"][k])
                    for k in self.map_at_k:
                        row.append(scores[nm]["map@k"][k])
                w.writerow(row)
        if self.main_score_function is None:
            return max(scores[nm]["map@k"][max(self.map_at_k)] for nm in self.score_function_names)
        return scores[self.main_score_function]["map@k"][max(self.map_at_k)]

    def _embed(self, model, texts):
        return model.encode(texts, batch_size=self.batch_size, show_progress_bar=self.show_progress_bar,
                            convert_to_tensor=True)

    def compute_metrices(self, model, corpus_model=None, corpus_embeddings=None) -> Dict[str, dict]:
        import torch
        from . import util
        if corpus_model is None:
            corpus_model = model
        max_k = max(max(self.mrr_at_k), max(self.ndcg_at_k), max(self.accuracy_at_k), max(self.precision_recall_at_k),
                    max(self.map_at_k))
        q_emb = self._embed(model, self.queries)
        if self._resolved is None:
            self._resolved = {nm: resolve_score_function(nm, self.score_functions[nm], q_emb.device)
                              for nm in self.score_function_names}
        best = {nm: None for nm in self.score_function_names}        # name -> (scores [nq, <=max_k], corpus rows)
        for start in range(0, len(self.corpus), self.corpus_chunk_size):
            end = min(start + self.corpus_chunk_size, len(self.corpus))
            c_emb = corpus_embeddings[start:end] if corpus_embeddings is not None else \
                self._embed(corpus_model, self.corpus[start:end])
            c_emb = torch.as_tensor(c_emb).to(q_emb.device)
            k = min(max_k, end - start)
            for nm in self.score_function_names:
                kind, how = self._resolved[nm]
                if kind == "native":
                    sc, idx = util.topk_scores(q_emb, c_emb, k, mode=how)
                else:                                                # foreign callable: its matrix, our selection
                    full = torch.as_tensor(how(q_emb, c_emb)).to(q_emb.device, torch.float32)
                    if tuple(full.shape) != (q_emb.shape[0], c_emb.shape[0]):
                        raise ValueError(f"score function {nm!r} returned shape {tuple(full.shape)}, expected "
                                         f"{(q_emb.shape[0], c_emb.shape[0])}")
                    sc, idx = util.topk_rows(full, k)
                idx = idx + start
                if best[nm] is not None:                             # merge with the hits of earlier chunks
                    sc = torch.cat([best[nm][0], sc], dim=1)
                    idx = torch.cat([best[nm][1], idx], dim=1)
                    sc, idx = util.topk_rows(sc, min(max_k, sc.shape[1]), index_map=idx)
                best[nm] = (sc, idx)
        out = {}
        for nm in self.score_function_names:
            sc, idx = best[nm][0].cpu().numpy(), best[nm][1].cpu().numpy()
            results = [[{"corpus_id": self.corpus_ids[int(c)], "score": float(s)} for s, c in zip(sc[qi], idx[qi])]
                       for qi in range(len(self.queries_ids))]
            out[nm] = ir_metrics(results, self.queries_ids, self.relevant_docs, self.mrr_at_k, self.ndcg_at_k,
                                 self.accuracy_at_k, self.precision_recall_at_k, self.map_at_k)
        return out
