"""GPU: retrieval scoring + top-k kernels (libqst qst_topk_rows / qst_topk_scores) against the fp64 oracle, and the
InformationRetrievalEvaluator end to end (encode -> score -> top-k across corpus chunks -> metrics) against the same
metrics computed by the oracle from the model's own embeddings."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib, util  # noqa: E402
from quadruplet_sentence_transformer_amd.evaluation import InformationRetrievalEvaluator  # noqa: E402
from quadruplet_sentence_transformer_amd.sentence_transformer import SentenceTransformer  # noqa: E402
from oracle import ir_oracle  # noqa: E402


@pytest.mark.parametrize("rows,n,k", [(1, 1, 1), (3, 10, 1), (5, 100, 100), (4, 1000, 10), (2, 50000, 100),
                                      (3, 5000, 1024), (7, 257, 33)])
def test_topk_rows_matches_a_full_sort(rows, n, k):
    g = torch.Generator().manual_seed(rows * 1000 + n + k)
    s = torch.randn(rows, n, generator=g)
    if n >= 100:
        s[:, ::7] = s[:, 3:4]                         # many exact ties
        s[0, 5] = float("inf")
        s[-1, 11] = -float("inf")
    sd = s.cuda()
    vals, idx = util.topk_rows(sd, k)
    order = np.lexsort((np.broadcast_to(np.arange(n), (rows, n)), -s.numpy().astype(np.float64)), axis=1)[:, :k]
    want = np.take_along_axis(s.numpy(), order, axis=1)
    np.testing.assert_array_equal(vals.cpu().numpy(), want)                       # bit-exact values, sorted
    got_idx = idx.cpu().numpy()
    np.testing.assert_array_equal(np.take_along_axis(s.numpy(), got_idx, axis=1), want)   # indices point at them
    for r in range(rows):
        assert len(set(got_idx[r].tolist())) == k                                 # no element taken twice
        strictly = want[r] > want[r][-1]                                          # above the cut: the choice is forced
        np.testing.assert_array_equal(got_idx[r][strictly], order[r][strictly])
    # index_map translates columns into caller ids
    imap = torch.arange(n, dtype=torch.int64).flip(0).repeat(rows, 1).cuda() + 1000
    vals2, idx2 = util.topk_rows(sd, k, index_map=imap)
    np.testing.assert_array_equal(vals2.cpu().numpy(), want)
    np.testing.assert_array_equal(np.take_along_axis(s.numpy(), (n - 1) - (idx2.cpu().numpy() - 1000), axis=1), want)


def test_topk_rows_bad_arguments():
    lib = _lib.load()
    s = torch.zeros(2, 8, device="cuda")
    o, i = torch.empty(2, 9, device="cuda"), torch.empty(2, 9, dtype=torch.int64, device="cuda")
    st = _lib.current_stream_ptr()
    assert lib.qst_topk_rows(s.data_ptr(), 8, None, 2, 8, 9, o.data_ptr(), i.data_ptr(), st) == -2      # k > n
    assert lib.qst_topk_rows(s.data_ptr(), 4, None, 2, 8, 2, o.data_ptr(), i.data_ptr(), st) == -1      # ld < n
    assert lib.qst_topk_rows(None, 8, None, 2, 8, 2, o.data_ptr(), i.data_ptr(), st) == -1


@pytest.mark.parametrize("nq,nc,dim,k,cosine", [(5, 37, 64, 10, True), (33, 1001, 384, 100, True), (7, 300, 128, 5, False),
                                                (2100, 2500, 64, 3, True), (1, 64, 768, 64, False)])
def test_topk_scores_matches_fp64_ranking(nq, nc, dim, k, cosine):
    g = torch.Generator().manual_seed(nq + nc + dim)
    q = torch.randn(nq, dim, generator=g)
    c = torch.randn(nc, dim, generator=g) * (0.5 + torch.rand(nc, 1, generator=g))     # varied norms
    vals, idx = util.topk_scores(q.cuda(), c.cuda(), k, cosine=cosine)
    want_s, want_i = ir_oracle.rank(q.numpy(), c.numpy(), k, cosine)
    scale = 1.0 if cosine else float(np.abs(want_s).max())
    np.testing.assert_allclose(vals.cpu().numpy(), want_s, rtol=0, atol=2e-5 * scale)     # split-bf16 x3 products
    got_i = idx.cpu().numpy()
    # the same documents, except where two candidates are closer than the arithmetic can tell apart
    gap_ok = 0
    for r in range(nq):
        if np.array_equal(got_i[r], want_i[r]):
            continue
        full = ir_oracle.rank(q.numpy()[r:r + 1], c.numpy(), nc, cosine)[0][0]
        pos = {int(d): p for p, d in enumerate(ir_oracle.rank(q.numpy()[r:r + 1], c.numpy(), nc, cosine)[1][0])}
        for a, b in zip(got_i[r], want_i[r]):
            if a != b:
                assert abs(full[pos[int(a)]] - full[pos[int(b)]]) < 4e-5 * scale
                gap_ok += 1
    assert gap_ok <= max(2, nq * k // 200)


@pytest.mark.parametrize("nq,nc,dim,k", [(5, 37, 64, 10), (33, 1001, 384, 100), (130, 300, 96, 7), (1, 64, 768, 64)])
def test_topk_scores_euclid_matches_fp64_ranking(nq, nc, dim, k):
    """mode 'euclid' = the reference's euclidean_score (models/evaluators.py:392-405). Near-duplicates of the queries are
    planted in the corpus: their distances (1e-3 .. 1e-1 at norm ~ sqrt(dim)) are what a |q|^2+|c|^2-2qc formulation
    would lose; the direct-difference kernel must rank them as float64 does."""
    g = torch.Generator().manual_seed(nq * 7 + nc + dim)
    q = torch.randn(nq, dim, generator=g)
    c = torch.randn(nc, dim, generator=g) * (0.5 + torch.rand(nc, 1, generator=g))
    ndup = min(nq, nc // 3, 8)
    for t in range(ndup):
        c[3 * t] = q[t] + 10.0 ** (-3 + 2 * t / max(1, ndup - 1)) * torch.randn(dim, generator=g) / dim ** 0.5
    vals, idx = util.topk_scores(q.cuda(), c.cuda(), k, mode="euclid")
    want_s, want_i = ir_oracle.rank(q.numpy(), c.numpy(), k, "euclid")
    np.testing.assert_allclose(vals.cpu().numpy(), want_s, rtol=2e-6, atol=1e-7)      # fp32 FMA chain + sqrt + rcp
    got_i = idx.cpu().numpy()
    full = ir_oracle.scores(q.numpy(), c.numpy(), "euclid")
    for r in range(nq):
        for a, b in zip(got_i[r], want_i[r]):
            assert a == b or abs(full[r, a] - full[r, b]) < 1e-6, (r, a, b)
    for t in range(ndup):
        assert got_i[t, 0] == 3 * t                                                    # the planted near-duplicate wins


@pytest.mark.parametrize("nq,nc,dim", [(3, 5, 32), (17, 130, 384), (2, 9, 100)])
def test_score_functions_match_fp64(nq, nc, dim):
    """util.cos_sim / dot_score / euclidean_score as functions (qst_score_matrix); dim 100 exercises the zero padding."""
    g = torch.Generator().manual_seed(nq + nc + dim)
    q = torch.randn(nq, dim, generator=g) * 1.3
    c = torch.randn(nc, dim, generator=g) * 0.7 + 0.1
    for fn, mode, tol in ((util.cos_sim, "cos", 2e-5), (util.dot_score, "dot", None), (util.euclidean_score, "euclid", 1e-6)):
        want = ir_oracle.scores(q.numpy(), c.numpy(), mode)
        got = fn(q.cuda(), c.cuda())
        assert got.is_cuda and tuple(got.shape) == (nq, nc)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=tol if tol else 2e-5 * float(np.abs(want).max()))
        host = fn(q, c)                                     # host tensors in -> host tensor out, same arithmetic
        assert not host.is_cuda and torch.equal(host, got.cpu())
    one = util.cos_sim(q[0], c[1])                          # 1-D inputs are rows, as in sentence_transformers.util
    assert tuple(one.shape) == (1, 1)


class _HashTexts:
    """Deterministic pseudo-sentences over a small vocabulary (the synthetic tokenizer hashes words to ids)."""
    words = ("a man rides red horse two dogs play in park woman eats green apple near old bridge small cat sleeps "
             "quick brown fox jumps over lazy river stone tower bright morning").split()

    @classmethod
    def make(cls, seed, n):
        rng = np.random.RandomState(seed)
        return " ".join(rng.choice(cls.words, size=n))


@pytest.mark.parametrize("preset,chunk", [("tiny-bert", 50000), ("tiny-bert", 17), ("tiny-mpnet", 40)])
def test_ir_evaluator_end_to_end(tmp_path, preset, chunk):
    model = SentenceTransformer(preset, device="cuda")
    corpus = {f"d{i}": _HashTexts.make(i, 5 + i % 9) for i in range(90)}
    queries, relevant = {}, {}
    rng = np.random.RandomState(7)
    for qn in range(23):
        base = rng.randint(0, 90)
        queries[f"q{qn}"] = corpus[f"d{base}"] + " " + _HashTexts.make(1000 + qn, 2)     # a perturbed corpus sentence
        relevant[f"q{qn}"] = {f"d{base}", f"d{(base + 1) % 90}"}
    queries["unused"] = "no relevant documents"                                          # dropped by the constructor
    ks = dict(mrr_at_k=[10], ndcg_at_k=[10], accuracy_at_k=[1, 3, 5, 10], precision_recall_at_k=[1, 3, 5, 10], map_at_k=[30])
    ev = InformationRetrievalEvaluator(queries, corpus, relevant, corpus_chunk_size=chunk, name="t", batch_size=16, **ks)
    got = ev.compute_metrices(model)
    # oracle: the same embeddings (fp32 from encode), ranked and scored on the CPU in float64
    q_emb = model.encode(ev.queries, batch_size=16, convert_to_numpy=True)
    c_emb = model.encode(ev.corpus, batch_size=16, convert_to_numpy=True)
    for name, cosine in (("cos_sim", True), ("dot_score", False)):
        _, order = ir_oracle.rank(q_emb, c_emb, 30, cosine)
        ranked = [[ev.corpus_ids[j] for j in row] for row in order]
        want = ir_oracle.metrics(ranked, [relevant[q] for q in ev.queries_ids], ks["mrr_at_k"], ks["ndcg_at_k"],
                                 ks["accuracy_at_k"], ks["precision_recall_at_k"], ks["map_at_k"])
        for metric in want:
            for k in want[metric]:
                assert got[name][metric][k] == pytest.approx(want[metric][k], abs=1e-9), (name, metric, k)
    score = ev(model, output_path=str(tmp_path), epoch=1, steps=2)
    assert score == pytest.approx(max(got[n]["map@k"][30] for n in ("cos_sim", "dot_score")), abs=1e-12)
    rows = open(os.path.join(str(tmp_path), ev.csv_file)).read().strip().splitlines()
    assert len(rows) == 2 and rows[0].split(",") == ev.csv_headers and rows[1].startswith("1,2,")


def _reference_euclidean_score(a, b):
    """models/evaluators.py:392-405 restated (the reference module cannot be imported offline): a foreign callable."""
    a = a if isinstance(a, torch.Tensor) else torch.tensor(a)
    b = b if isinstance(b, torch.Tensor) else torch.tensor(b)
    a = a.unsqueeze(0) if a.dim() == 1 else a
    b = b.unsqueeze(0) if b.dim() == 1 else b
    return 1 / (1 + torch.cdist(a, b, p=2))


def test_score_function_resolution_is_by_behaviour():
    from quadruplet_sentence_transformer_amd.evaluation import resolve_score_function
    assert resolve_score_function("euclid_score", _reference_euclidean_score, "cuda") == ("native", 2)
    assert resolve_score_function("whatever", lambda a, b: a @ b.T, "cuda") == ("native", 0)
    custom = lambda a, b: -torch.cdist(a, b, p=1)                                  # Manhattan: none of the native modes
    assert resolve_score_function("cos_sim", custom, "cuda") == ("callable", custom)    # the NAME does not decide
    cpu_only = lambda a, b: torch.tensor(np.asarray(a.cpu()) @ np.asarray(b.cpu()).T)   # returns a host tensor
    assert resolve_score_function("np_dot", cpu_only, "cuda") == ("native", 0)
    broken = lambda a, b: (_ for _ in ()).throw(RuntimeError("no probe"))
    assert resolve_score_function("x", broken, "cuda") == ("callable", broken)


@pytest.mark.parametrize("chunk", [50000, 17])
def test_ir_evaluator_with_the_reference_score_functions(tmp_path, chunk):
    """Exactly the dictionary training/main.py:57 / ir_evauation_script.py:71 build, through the constructor call of
    models/evaluators.py:572-588, on one and on several corpus chunks; plus a custom callable under a stock name."""
    model = SentenceTransformer("tiny-bert", device="cuda")
    corpus = {f"d{i}": _HashTexts.make(i, 5 + i % 9) for i in range(90)}
    queries, relevant = {}, {}
    rng = np.random.RandomState(5)
    for qn in range(19):
        base = rng.randint(0, 90)
        queries[f"q{qn}"] = corpus[f"d{base}"] + " " + _HashTexts.make(2000 + qn, 2)
        relevant[f"q{qn}"] = {f"d{base}", f"d{(base + 3) % 90}"}
    manhattan = lambda a, b: -torch.cdist(a, b, p=1)
    score_functions = {"cos_sim": util.cos_sim, "dot_score": util.dot_score, "euclid_score": _reference_euclidean_score,
                       "manhattan": manhattan}
    ks = dict(mrr_at_k=[10], ndcg_at_k=[10], accuracy_at_k=[1, 3, 5, 10], precision_recall_at_k=[1, 3, 5, 10], map_at_k=[100])
    ev = InformationRetrievalEvaluator(queries=queries, corpus=corpus, relevant_docs=relevant, corpus_chunk_size=chunk,
                                       show_progress_bar=False, batch_size=32, name="exp", write_csv=True,
                                       score_functions=score_functions, main_score_function=None, **ks)
    got = ev.compute_metrices(model)
    assert ev._resolved["euclid_score"] == ("native", 2) and ev._resolved["manhattan"][0] == "callable"
    q_emb = model.encode(ev.queries, batch_size=32, convert_to_numpy=True)
    c_emb = model.encode(ev.corpus, batch_size=32, convert_to_numpy=True)
    man = -np.abs(q_emb.astype(np.float64)[:, None, :] - c_emb.astype(np.float64)[None, :, :]).sum(-1)
    for name, mode in (("cos_sim", "cos"), ("dot_score", "dot"), ("euclid_score", "euclid"), ("manhattan", None)):
        s = man if mode is None else ir_oracle.scores(q_emb, c_emb, mode)
        order = np.lexsort((np.broadcast_to(np.arange(s.shape[1]), s.shape), -s), axis=1)[:, :90]
        ranked = [[ev.corpus_ids[j] for j in row] for row in order]
        want = ir_oracle.metrics(ranked, [relevant[q] for q in ev.queries_ids], ks["mrr_at_k"], ks["ndcg_at_k"],
                                 ks["accuracy_at_k"], ks["precision_recall_at_k"], ks["map_at_k"])
        for metric in want:
            for k in want[metric]:
                assert got[name][metric][k] == pytest.approx(want[metric][k], abs=1e-9), (name, metric, k)
    score = ev(model, output_path=str(tmp_path), epoch=0, steps=5)
    assert score == pytest.approx(max(got[n]["map@k"][100] for n in score_functions), abs=1e-12)
    rows = open(os.path.join(str(tmp_path), ev.csv_file)).read().strip().splitlines()
    assert rows[0].split(",") == ev.csv_headers and "euclid_score-MAP@100" in rows[0]


def test_hard_negative_mining_matches_brute_force():
    """for every reference the k most similar candidates among those with cosine <= threshold
    (the reference's NEG_EXAMPLE_SIM_TRESHOLD filter + hard_contrastive_sampling), all references in one pass."""
    g = torch.Generator().manual_seed(11)
    R_, C_, D, k, thr = 37, 500, 128, 6, 0.05
    q = torch.randn(R_, D, generator=g)
    c = torch.randn(C_, D, generator=g)
    c[:40] = q[:1] + 0.3 * torch.randn(40, D, generator=g)          # near-duplicates of reference 0: must be filtered
    idx, sc = util.mine_hard_negatives(q.cuda(), c.cuda(), k, threshold=thr)
    qn = (q / q.norm(dim=1, keepdim=True)).double()
    cn = (c / c.norm(dim=1, keepdim=True)).double()
    s = (qn @ cn.T).numpy()
    for r in range(R_):
        ok = np.where(s[r] <= thr)[0]
        want = ok[np.argsort(-s[r][ok], kind="stable")][:k]
        got = idx[r].cpu().numpy()
        np.testing.assert_allclose(sc[r].cpu().numpy()[:len(want)], s[r][want], rtol=0, atol=2e-5)
        assert float(sc[r].max()) <= thr + 1e-6
        assert set(got[:len(want)].tolist()) == set(want.tolist()) or np.allclose(s[r][got[:len(want)]], s[r][want], atol=4e-5)
    assert not (set(idx[0].cpu().tolist()) & set(range(40)))       # the paraphrase-like candidates never come back
    # fewer qualifying candidates than k -> padded with (-1, -inf)
    idx2, sc2 = util.mine_hard_negatives(q[:2].cuda(), c[:3].cuda(), 5, threshold=-0.99)
    assert (idx2 == -1).all() and torch.isinf(sc2).all()
    # sentences + embedder
    model = SentenceTransformer("tiny-bert", device="cuda")
    refs = [_HashTexts.make(i, 7) for i in range(5)]
    cands = [_HashTexts.make(100 + i, 6) for i in range(30)]
    i3, s3 = util.mine_hard_negatives(refs, cands, 4, threshold=0.9, embedder=model)
    e_r, e_c = model.encode(refs, convert_to_tensor=True), model.encode(cands, convert_to_tensor=True)
    i4, s4 = util.mine_hard_negatives(e_r, e_c, 4, threshold=0.9)
    assert torch.equal(i3, i4) and torch.allclose(s3, s4)
    with pytest.raises(ValueError):
        util.mine_hard_negatives(refs, cands, 4)
