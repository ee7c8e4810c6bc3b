"""sentence_transformers.util functions the reference imports (training/main.py:6, models/evaluators.py:9-12):
cos_sim, dot_score, batch_to_device. Tiny host-side helpers over torch tensors, off the throughput path."""
from __future__ import annotations

import numpy as np
import torch

from .sentence_transformer import batch_to_device  # noqa: F401


def _as_2d(x):
    if not isinstance(x, torch.Tensor):
        x = torch.tensor(np.asarray(x))
    if x.dim() == 1:
        x = x.unsqueeze(0)
    return x


def cos_sim(a, b) -> torch.Tensor:
    a, b = _as_2d(a), _as_2d(b)
    a = torch.nn.functional.normalize(a, p=2, dim=1)
    b = torch.nn.functional.normalize(b, p=2, dim=1)
    return torch.mm(a, b.transpose(0, 1))


pytorch_cos_sim = cos_sim


def dot_score(a, b) -> torch.Tensor:
    a, b = _as_2d(a), _as_2d(b)
    return torch.mm(a, b.transpose(0, 1))


def topk_scores(queries: torch.Tensor, corpus: torch.Tensor, k: int, cosine: bool = True):
    """The k best corpus rows for every query row on the GPU (libqst qst_topk_scores: normalise / copy, split-bf16 x3
    matmul, radix-select top-k): what InformationRetrievalEvaluator does per corpus chunk with cos_sim/dot_score +
    torch.topk. Returns (scores [nq, k] descending, indices int64 [nq, k]). No CPU fallback."""
    from . import _lib
    lib = _lib.load()
    if not (queries.is_cuda and corpus.is_cuda):
        raise _lib.QstError("topk_scores runs on the HIP device: pass CUDA tensors (there is no CPU fallback)")
    q = queries.to(torch.float32).contiguous()
    c = corpus.to(torch.float32).contiguous()
    nq, dim = q.shape
    nc = c.shape[0]
    ws = torch.empty(lib.qst_topk_workspace_bytes(nq, nc, dim), dtype=torch.uint8, device=q.device)
    out_s = torch.empty(nq, k, dtype=torch.float32, device=q.device)
    out_i = torch.empty(nq, k, dtype=torch.int64, device=q.device)
    _lib.check(lib.qst_topk_scores(q.data_ptr(), c.data_ptr(), nq, nc, dim, k, int(cosine), out_s.data_ptr(),
                                   out_i.data_ptr(), ws.data_ptr(), ws.numel(), _lib.current_stream_ptr()),
               "qst_topk_scores")
    return out_s, out_i


def topk_rows(scores: torch.Tensor, k: int, index_map: torch.Tensor = None):
    """k best entries per row of a score matrix (libqst qst_topk_rows); index_map translates columns to ids."""
    from . import _lib
    lib = _lib.load()
    s = scores.to(torch.float32).contiguous()
    im = None if index_map is None else index_map.to(torch.int64).contiguous()
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
    q, c = q.to(torch.float32).contiguous(), c.to(torch.float32).contiguous()
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
