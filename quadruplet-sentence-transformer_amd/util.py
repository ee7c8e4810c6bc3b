"""sentence_transformers.util functions the reference imports (training/main.py:6, models/evaluators.py:9-12):
cos_sim, dot_score, batch_to_device -- plus euclidean_score, the reference's own third score function
(models/evaluators.py:392-405). All three score matrices come from libqst (qst_score_matrix: row normalisation, the
split-bf16 x3 GEMM or the direct-difference Euclidean kernel); there is no torch arithmetic and no CPU path here."""
from __future__ import annotations

import numpy as np
import torch

from .sentence_transformer import batch_to_device  # noqa: F401

SCORE_DOT, SCORE_COS, SCORE_EUCLID = 0, 1, 2
_MODES = {"dot": SCORE_DOT, "dot_score": SCORE_DOT, "cos": SCORE_COS, "cos_sim": SCORE_COS, "cosine": SCORE_COS,
          "euclid": SCORE_EUCLID, "euclid_score": SCORE_EUCLID, "euclidean_score": SCORE_EUCLID}


def _mode(m) -> int:
    if isinstance(m, str):
        return _MODES[m]
    if isinstance(m, bool):
        return SCORE_COS if m else SCORE_DOT
    return int(m)


def _as_2d(x):
    if not isinstance(x, torch.Tensor):
        x = torch.tensor(np.asarray(x))
    if x.dim() == 1:
        x = x.unsqueeze(0)
    return x


def _device_rows(x, dev=None):
    """fp32 contiguous rows on the HIP device, feature dimension zero-padded to a multiple of 32 (zeros change none of
    the three scores). Host tensors are moved: the arithmetic runs in libqst either way."""
    from . import _lib
    x = _as_2d(x)
    if not x.is_cuda:
        if not torch.cuda.is_available():
            raise _lib.QstError("score functions run on the HIP device and none is visible (there is no CPU path)")
        x = x.to(dev if dev is not None else torch.device("cuda", torch.cuda.current_device()))
    x = x.to(torch.float32)
    pad = (-x.shape[1]) % 32
    if pad:
        x = torch.nn.functional.pad(x, (0, pad))
    return x.contiguous()


def score_matrix(a, b, mode) -> torch.Tensor:
    """[len(a), len(b)] scores: mode 'dot' | 'cos' | 'euclid' (libqst qst_score_matrix). The result lives where `a` did."""
    from . import _lib
    lib = _lib.load()
    a0 = _as_2d(a)
    q = _device_rows(a0)
    c = _device_rows(b, q.device)
    if c.device != q.device:
        c = c.to(q.device)
    if q.shape[1] != c.shape[1]:
        raise ValueError(f"embedding sizes differ: {tuple(_as_2d(a).shape)} vs {tuple(_as_2d(b).shape)}")
    nq, dim = q.shape
    nc = c.shape[0]
    ld = (nc + 3) // 4 * 4
    with torch.cuda.device(q.device):
        out = torch.empty(nq, ld, dtype=torch.float32, device=q.device)
        ws = torch.empty(lib.qst_score_workspace_bytes(nq, nc, dim), dtype=torch.uint8, device=q.device)
        _lib.check(lib.qst_score_matrix(q.data_ptr(), c.data_ptr(), nq, nc, dim, _mode(mode), out.data_ptr(), ld,
                                        ws.data_ptr(), ws.numel(), _lib.current_stream_ptr()), "qst_score_matrix")
    out = out[:, :nc]
    return out if a0.is_cuda else out.cpu()


def cos_sim(a, b) -> torch.Tensor:
    return score_matrix(a, b, SCORE_COS)


pytorch_cos_sim = cos_sim


def dot_score(a, b) -> torch.Tensor:
    return score_matrix(a, b, SCORE_DOT)


def euclidean_score(a, b) -> torch.Tensor:
    """1 / (1 + ||a_i - b_j||_2): /root/reference/models/evaluators.py:392-405."""
    return score_matrix(a, b, SCORE_EUCLID)


# marks this package's own score functions: evaluators route them (and callables that behave like them) to the fused
# score + top-k kernel instead of materialising the matrix
cos_sim._qst_mode = SCORE_COS
dot_score._qst_mode = SCORE_DOT
euclidean_score._qst_mode = SCORE_EUCLID


def topk_scores(queries: torch.Tensor, corpus: torch.Tensor, k: int, cosine=True, mode=None):
    """The k best corpus rows for every query row on the GPU (libqst qst_topk_scores: normalise / copy, split-bf16 x3
    matmul or Euclidean kernel, radix-select top-k): what InformationRetrievalEvaluator does per corpus chunk with its
    score function + torch.topk. mode: 'dot' | 'cos' | 'euclid' (`cosine` is the older boolean spelling).
    Returns (scores [nq, k] descending, indices int64 [nq, k]). No CPU fallback."""
    from . import _lib
    lib = _lib.load()
    if not (queries.is_cuda and corpus.is_cuda):
        raise _lib.QstError("topk_scores runs on the HIP device: pass CUDA tensors (there is no CPU fallback)")
    m = _mode(mode if mode is not None else cosine)
    q = _device_rows(queries)
    c = _device_rows(corpus, q.device)
    nq, dim = q.shape
    nc = c.shape[0]
    ws = torch.empty(lib.qst_topk_workspace_bytes(nq, nc, dim), dtype=torch.uint8, device=q.device)
    out_s = torch.empty(nq, k, dtype=torch.float32, device=q.device)
    out_i = torch.empty(nq, k, dtype=torch.int64, device=q.device)
    _lib.check(lib.qst_topk_scores(q.data_ptr(), c.data_ptr(), nq, nc, dim, k, m, out_s.data_ptr(),
                                   out_i.data_ptr(), ws.data_ptr(), ws.numel(), _lib.current_stream_ptr()),
               "qst_topk_scores")
    return out_s, out_i


def topk_rows(scores: torch.Tensor, k: int, index_map: torch.Tensor = None):
    """k best entries per row of a score matrix (libqst qst_topk_rows); index_map translates columns to ids."""
    from . import _lib
    lib = _lib.load()
    if not scores.is_cuda:
        raise _lib.QstError("topk_rows runs on the HIP device: pass a CUDA tensor (there is no CPU fallback)")
    s = scores.to(torch.float32).contiguous()
    im = None if index_map is None else index_map.to(s.device, torch.int64).contiguous()
    n_rows, n = s.shape
    out_s = torch.empty(n_rows, k, dtype=torch.float32, device=s.device)
    out_i = torch.empty(n_rows, k, dtype=torch.int64, device=s.device)
    _lib.check(lib.qst_topk_rows(s.data_ptr(), n, _lib.ptr(im), n_rows, n, k, out_s.data_ptr(), out_i.data_ptr(),
                                 _lib.current_stream_ptr()), "qst_topk_rows")
    return out_s, out_i


def mine_hard_negatives(references, candidates, k: int, threshold: float = 0.2, embedder=None, batch_size: int = 64):
    """Negative selection of the reference's dataset (dataset/quadruplet_dataset.py:185-270) for MANY reference
    captions in one GPU pass: among `candidates`, those whose cosine similarity to a reference is <= threshold
    (NEG_EXAMPLE_SIM_TRESHOLD = 0.2, the "not a paraphrase" filter), and of those the k most similar
    (hard_contrastive_sampling(max_mode=True), :31-47). The reference does this per item with two encode() calls on
    the training thread; here both sides are encoded in batches and scored by libqst (qst_topk_scores_capped).

    references / candidates: lists of str (then `embedder` -- a SentenceTransformer -- encodes them) or CUDA tensors of
    embeddings [R, D] / [C, D]. Returns (index int64 [R, k] into candidates, -1 where fewer than k qualify;
    score f32 [R, k], -inf there)."""
    from . import _lib
    lib = _lib.load()

    def emb(x):
        if torch.is_tensor(x):
            return x
        if embedder is None:
            raise ValueError("pass embeddings, or sentences together with an embedder")
        return embedder.encode(list(x), batch_size=batch_size, convert_to_tensor=True)
    q, c = emb(references), emb(candidates)
    if not (q.is_cuda and c.is_cuda):
        raise _lib.QstError("mine_hard_negatives runs on the HIP device: embeddings must be CUDA tensors")
    q = _device_rows(q)
    c = _device_rows(c, q.device)
    nq, dim = q.shape
    nc = c.shape[0]
    kk = min(k, nc)
    ws = torch.empty(lib.qst_topk_workspace_bytes(nq, nc, dim), dtype=torch.uint8, device=q.device)
    out_s = torch.full((nq, k), float("-inf"), dtype=torch.float32, device=q.device)
    out_i = torch.full((nq, k), -1, dtype=torch.int64, device=q.device)
    ts = torch.empty(nq, kk, dtype=torch.float32, device=q.device)
    ti = torch.empty(nq, kk, dtype=torch.int64, device=q.device)
    _lib.check(lib.qst_topk_scores_capped(q.data_ptr(), c.data_ptr(), nq, nc, dim, kk, 1, float(threshold), ts.data_ptr(),
                                          ti.data_ptr(), ws.data_ptr(), ws.numel(), _lib.current_stream_ptr()),
               "qst_topk_scores_capped")
    out_s[:, :kk], out_i[:, :kk] = ts, ti
    return out_i, out_s
