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
        InformationRetrievalEvaluator(queries, corpus, rel, score_functions={"manhattan": None})     # no callable, unknown name
    with pytest.raises(ValueError):
        InformationRetrievalEvaluator(queries, corpus, rel, score_functions={"cos_sim": 3})          # not callable


def _reference_euclidean_score(a, b):
    """What the reference defines at models/evaluators.py:392-405 (restated: the reference module cannot be imported
    offline) -- a FOREIGN callable as far as this package is concerned."""
    import torch
    a = a if isinstance(a, torch.Tensor) else torch.tensor(a)
    b = b if isinstance(b, torch.Tensor) else torch.tensor(b)
    a = a.unsqueeze(0) if a.dim() == 1 else a
    b = b.unsqueeze(0) if b.dim() == 1 else b
    return 1 / (1 + torch.cdist(a, b, p=2))


def test_evaluator_takes_the_score_functions_the_reference_passes():
    """training/main.py:57 and ir_evauation_script.py:71 always build this dictionary; main.py:74-93 /
    evaluators.py:572-588 forward it with exactly these keyword arguments."""
    from quadruplet_sentence_transformer_amd import util
    from quadruplet_sentence_transformer_amd.evaluation import resolve_score_function
    score_functions = {"cos_sim": util.cos_sim, "dot_score": util.dot_score, "euclid_score": _reference_euclidean_score}
    ev = InformationRetrievalEvaluator(
        queries={"q1": "a"}, corpus={"d1": "x", "d2": "y"}, relevant_docs={"q1": {"d1"}}, corpus_chunk_size=50000,
        mrr_at_k=[10], ndcg_at_k=[10], accuracy_at_k=[1, 3, 5, 10], precision_recall_at_k=[1, 3, 5, 10], map_at_k=[100],
        show_progress_bar=False, batch_size=32, name="exp", write_csv=True, score_functions=score_functions,
        main_score_function=None)
    assert ev.score_function_names == ["cos_sim", "dot_score", "euclid_score"]
    assert "euclid_score-MAP@100" in ev.csv_headers and len(ev.csv_headers) == 2 + 3 * 15
    # this package's own functions are native by their tag, whatever name they are registered under; None needs a known name
    assert resolve_score_function("anything", util.euclidean_score) == ("native", 2)
    assert resolve_score_function("cos_sim", None) == ("native", 1)
    assert resolve_score_function("euclid_score", None) == ("native", 2)
    with pytest.raises(ValueError):
        resolve_score_function("manhattan", None)


def test_fit_accepts_the_reference_keyword_set():
    """training/main.py:128-148 verbatim: every keyword binds to the drop-in's fit()."""
    import inspect
    import torch
    from quadruplet_sentence_transformer_amd.sentence_transformer import SentenceTransformer
    kwargs = dict(train_objectives=[(None, None)], evaluator=None, epochs=1, steps_per_epoch=None, scheduler="WarmupLinear",
                  warmup_steps=100, optimizer_class=torch.optim.AdamW, optimizer_params={"lr": 2e-5}, weight_decay=0.01,
                  evaluation_steps=0, output_path="out", save_best_model=True, max_grad_norm=1.0, use_amp=False,
                  callback=None, show_progress_bar=False, checkpoint_path="ckpt", checkpoint_save_steps=500,
                  checkpoint_save_total_limit=0)
    bound = inspect.signature(SentenceTransformer.fit).bind(None, **kwargs)
    assert set(kwargs) <= set(bound.arguments)


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
    with pytest.raises(_lib.QstError):
        util.topk_rows(torch.zeros(2, 32), 1)
    if not torch.cuda.is_available():                  # score functions have no CPU arithmetic either
        with pytest.raises(_lib.QstError):
            util.cos_sim(torch.zeros(2, 32), torch.zeros(4, 32))
