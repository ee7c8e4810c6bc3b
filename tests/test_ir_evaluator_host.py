"""CPU: retrieval-metric arithmetic of the IR evaluator against a hand-worked example and the oracle restatement;
constructor behaviour (query filtering, CSV header layout, score-function validation); the drop-in namespace exports
what the reference's evaluator module imports (models/evaluators.py:9-12)."""
import os
import sys

import numpy as np
import pytest

import quadruplet_sentence_transformer_amd  # noqa: F401
from quadruplet_sentence_transformer_amd.evaluation import InformationRetrievalEvaluator, ir_metrics
from oracle import ir_oracle


def test_ir_metrics_hand_worked_example():
    # one query, relevant = {a, c, x}; ranking (best first) = a, b, c, d
    hits = [[{"corpus_id": "b", "score": 0.8}, {"corpus_id": "a", "score": 0.9}, {"corpus_id": "d", "score": 0.1},
             {"corpus_id": "c", "score": 0.5}]]
    m = ir_metrics(hits, ["q"], {"q": {"a", "c", "x"}}, [1, 4], [4], [1, 2], [2, 4], [4])
    assert m["accuracy@k"] == {1: 1.0, 2: 1.0}
    assert m["precision@k"][2] == pytest.approx(0.5) and m["precision@k"][4] == pytest.approx(0.5)
    assert m["recall@k"][2] == pytest.approx(1 / 3) and m["recall@k"][4] == pytest.approx(2 / 3)
    assert m["mrr@k"] == {1: 1.0, 4: 1.0}
    dcg = 1 / np.log2(2) + 1 / np.log2(4)
    idcg = 1 / np.log2(2) + 1 / np.log2(3) + 1 / np.log2(4)
    assert m["ndcg@k"][4] == pytest.approx(dcg / idcg)
    assert m["map@k"][4] == pytest.approx((1 / 1 + 2 / 3) / 3)          # divided by min(k, |relevant|) = 3


def test_ir_metrics_match_oracle_on_random_rankings():
    rng = np.random.default_rng(3)
    ids = [f"d{i}" for i in range(40)]
    qids, results, relevant = [], [], {}
    for q in range(25):
        qid = f"q{q}"
        qids.append(qid)
        perm = rng.permutation(40)[:20]
        scores = np.sort(rng.random(20))[::-1]
        order = rng.permutation(20)                                    # hit lists arrive unsorted
        results.append([{"corpus_id": ids[perm[i]], "score": float(scores[i])} for i in order])
        relevant[qid] = set(ids[j] for j in rng.choice(40, size=rng.integers(1, 6), replace=False))
    ks = dict(mrr_at_k=[1, 10], ndcg_at_k=[5, 10], accuracy_at_k=[1, 3, 10], precision_recall_at_k=[1, 5], map_at_k=[10, 20])
    got = ir_metrics(results, qids, relevant, **ks)
    ranked = [[h["corpus_id"] for h in sorted(r, key=lambda h: -h["score"])] for r in results]
    want = ir_oracle.metrics(ranked, [relevant[q] for q in qids], ks["mrr_at_k"], ks["ndcg_at_k"], ks["accuracy_at_k"],
                             ks["precision_recall_at_k"], ks["map_at_k"])
    for name in want:
        for k in want[name]:
            assert got[name][k] == pytest.approx(want[name][k], abs=1e-12), (name, k)


def test_evaluator_constructor_follows_st_conventions():
    queries = {"q1": "a", "q2": "b", "q3": "c"}
    corpus = {"d1": "x", "d2": "y"}
    rel = {"q1": {"d1"}, "q2": set(), "q4": {"d2"}}              # q2: nothing relevant, q3: no entry -> both dropped
    ev = InformationRetrievalEvaluator(queries, corpus, rel, name="val", mrr_at_k=[10], ndcg_at_k=[10], accuracy_at_k=[1],
                                       precision_recall_at_k=[1], map_at_k=[100])
    assert ev.queries_ids == ["q1"] and ev.queries == ["a"] and ev.corpus_ids == ["d1", "d2"]
    assert ev.csv_file == "Information-Retrieval_evaluation_val_results.csv"
    assert ev.score_function_names == ["cos_sim", "dot_score"]
    assert ev.csv_headers == ["epoch", "steps", "cos_sim-Accuracy@1", "cos_sim-Precision@1", "cos_sim-Recall@1",
                              "cos_sim-MRR@10", "cos_sim-NDCG@10", "cos_sim-MAP@100", "dot_score-Accuracy@1",
                              "dot_score-Precision@1", "dot_score-Recall@1", "dot_score-MRR@10", "dot_score-NDCG@10",
                              "dot_score-MAP@100"]
    with pytest.raises(ValueError):
        InformationRetrievalEvaluator(queries, corpus, rel, score_functions={"manhattan": None})


def test_dropin_namespace_exports_what_the_reference_imports():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "dropin"))
    try:
        for m in [k for k in sys.modules if k == "sentence_transformers" or k.startswith("sentence_transformers.")]:
            del sys.modules[m]
        import sentence_transformers as st
        from sentence_transformers.evaluation import (InformationRetrievalEvaluator as IRE, SentenceEvaluator,  # noqa: F401
                                                      SequentialEvaluator, SimilarityFunction, TripletEvaluator)
        from sentence_transformers.util import batch_to_device, cos_sim, dot_score  # noqa: F401
        assert IRE is InformationRetrievalEvaluator
        ce = st.CrossEncoder("cross-encoder/stsb-roberta-large")        # models/evaluators.py:31 runs this at import
        with pytest.raises(RuntimeError):
            ce.predict([("a", "b")])
    finally:
        sys.path.remove(os.path.join(root, "dropin"))
        for m in [k for k in sys.modules if k == "sentence_transformers" or k.startswith("sentence_transformers.")]:
            del sys.modules[m]


def test_topk_refuses_cpu_tensors():
    import torch
    from quadruplet_sentence_transformer_amd import _lib, util
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libqst.so not built")
    with pytest.raises(_lib.QstError):
        util.topk_scores(torch.zeros(2, 32), torch.zeros(4, 32), 1)
