"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the retrieval evaluator.

Restates, independently of quadruplet_sentence_transformer_amd.evaluation, what sentence-transformers 2.2.2's
InformationRetrievalEvaluator computes for the reference (call sites /root/reference/models/evaluators.py:572-588,
/root/reference/ir_evauation_script.py:107-131): util.cos_sim / util.dot_score of query vs corpus embeddings
(float64 here), the top max(k) hits per query, then Accuracy@k, Precision@k, Recall@k, MRR@k, NDCG@k (binary gain,
log2 discount, ideal = all relevant first), MAP@k (divided by min(k, |relevant|)). Ranking ties break by corpus
position, which is also what the HIP kernel does. Only tests/ may import this module."""
import math

import numpy as np


def scores(queries: np.ndarray, corpus: np.ndarray, mode) -> np.ndarray:
    """float64 score matrix. mode: True / 'cos' = util.cos_sim, False / 'dot' = util.dot_score, 'euclid' = the
    reference's euclidean_score (/root/reference/models/evaluators.py:392-405): 1 / (1 + cdist(a, b, p=2))."""
    q = np.asarray(queries, dtype=np.float64)
    c = np.asarray(corpus, dtype=np.float64)
    if mode == "euclid":
        d2 = ((q[:, None, :] - c[None, :, :]) ** 2).sum(-1) if q.shape[0] * c.shape[0] * q.shape[1] < 4e7 else \
            np.stack([((qi[None, :] - c) ** 2).sum(-1) for qi in q])
        return 1.0 / (1.0 + np.sqrt(d2))
    if mode is True or mode == "cos":
        q = q / np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-12)
        c = c / np.maximum(np.linalg.norm(c, axis=1, keepdims=True), 1e-12)
    return q @ c.T


def rank(queries: np.ndarray, corpus: np.ndarray, k: int, cosine):
    s = scores(queries, corpus, cosine)
    order = np.lexsort((np.broadcast_to(np.arange(s.shape[1]), s.shape), -s), axis=1)[:, :k]
    return np.take_along_axis(s, order, axis=1), order


def metrics(ranked_ids, relevant, ks_mrr, ks_ndcg, ks_acc, ks_pr, ks_map):
    """ranked_ids: per query, corpus ids best first; relevant: per query, set of relevant ids."""
    nq = len(ranked_ids)
    out = {"accuracy@k": {}, "precision@k": {}, "recall@k": {}, "mrr@k": {}, "ndcg@k": {}, "map@k": {}}
    for k in ks_acc:
        out["accuracy@k"][k] = sum(1 for r, rel in zip(ranked_ids, relevant) if set(r[:k]) & rel) / nq
    for k in ks_pr:
        out["precision@k"][k] = sum(len([d for d in r[:k] if d in rel]) / k for r, rel in zip(ranked_ids, relevant)) / nq
        out["recall@k"][k] = sum(len([d for d in r[:k] if d in rel]) / len(rel) for r, rel in zip(ranked_ids, relevant)) / nq
    for k in ks_mrr:
        tot = 0.0
        for r, rel in zip(ranked_ids, relevant):
            first = next((i for i, d in enumerate(r[:k]) if d in rel), None)
            tot += 0.0 if first is None else 1.0 / (first + 1)
        out["mrr@k"][k] = tot / nq
    for k in ks_ndcg:
        tot = 0.0
        for r, rel in zip(ranked_ids, relevant):
            dcg = sum(1.0 / math.log2(i + 2) for i, d in enumerate(r[:k]) if d in rel)
            idcg = sum(1.0 / math.log2(i + 2) for i in range(min(k, len(rel))))
            tot += dcg / idcg
        out["ndcg@k"][k] = tot / nq
    for k in ks_map:
        tot = 0.0
        for r, rel in zip(ranked_ids, relevant):
            hits, acc = 0, 0.0
            for i, d in enumerate(r[:k]):
                if d in rel:
                    hits += 1
                    acc += hits / (i + 1)
            tot += acc / min(k, len(rel))
        out["map@k"][k] = tot / nq
    return out
