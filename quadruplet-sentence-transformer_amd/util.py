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
