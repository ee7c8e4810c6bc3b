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
