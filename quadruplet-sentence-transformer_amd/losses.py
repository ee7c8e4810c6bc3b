"""Drop-in for the reference's `models.losses.losses` on the HIP path.

Same public surface as /root/reference/models/losses/losses.py: `gamma_quadruplet_loss` (:9-69),
`QuadrupletLoss` (:157-238), `GammaQuadrupletLoss` (:241-303), `DEFAULT_GAMMA`, `REDUCTIONS` -- same
argument names, defaults, property names and ValueError texts -- but the arithmetic (three
triplet_margin_loss terms and their autograd) is ONE fused HIP kernel, qst_quadruplet_loss
(csrc/loss.hip), reached through the C-ABI. Inputs must live on a HIP device: there is no CPU path here.
`d_regularized_quadruplet_loss` is out of scope (never wired into the reference's training).
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Optional

import torch

from . import _lib

DEFAULT_GAMMA = 0.6
REDUCTIONS = frozenset(["mean", "sum", "none"])
_RED_CODE = {"none": 0, "sum": 1, "mean": 2}


def _require_positive(name: str, value) -> None:
    if value <= 0:
        raise ValueError(f"{name} must be positive, {value} given")


def _require_gamma(gamma) -> None:
    if gamma < 0 or gamma > 1:
        raise ValueError(f"gamma must be between 0 and 1, {gamma} given")


def _require_reduction(reduction) -> None:
    if reduction not in REDUCTIONS:
        raise ValueError(f"reduction must be one of: {REDUCTIONS}, "
                         f"{reduction} given")


def _launch(xs, hp, red_code, grad_out, want_grads):
    lib = _lib.load()
    B, D = xs[0].shape
    dev = xs[0].device
    out = torch.empty(B if red_code == 0 else 1, dtype=torch.float32, device=dev)
    scratch = torch.empty(max(B, 1), dtype=torch.float32, device=dev)
    grads = [torch.empty_like(x) for x in xs] if want_grads else [None] * 4
    gamma, m_pn, m_pq, m_qn, p, swap = hp
    with torch.cuda.device(dev):
        _lib.check(lib.qst_quadruplet_loss(
            xs[0].data_ptr(), xs[1].data_ptr(), xs[2].data_ptr(), xs[3].data_ptr(), B, D,
            float(gamma), float(m_pn), float(m_pq), float(m_qn), float(p), int(bool(swap)), red_code,
            out.data_ptr(), _lib.ptr(grad_out), *[_lib.ptr(g) for g in grads], scratch.data_ptr(),
            _lib.current_stream_ptr()), "qst_quadruplet_loss")
    return out, grads


class _QuadrupletLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xa, xp, xq, xn, hp, red_code):
        xs = [x.detach().to(torch.float32).contiguous() for x in (xa, xp, xq, xn)]
        out, _ = _launch(xs, hp, red_code, None, False)
        ctx.save_for_backward(*xs)
        ctx.hp, ctx.red_code = hp, red_code
        ctx.in_dtypes = [x.dtype for x in (xa, xp, xq, xn)]
        return out if red_code == 0 else out.reshape(())

    @staticmethod
    def backward(ctx, grad_output):
        xs = list(ctx.saved_tensors)
        go = grad_output.detach().to(torch.float32).contiguous().reshape(-1)
        _, grads = _launch(xs, ctx.hp, ctx.red_code, go, True)      # the kernel recomputes the 6 distances (3 KB/row)
        grads = [g.to(dt) for g, dt in zip(grads, ctx.in_dtypes)]
        return grads[0], grads[1], grads[2], grads[3], None, None


def gamma_quadruplet_loss(x_anchor: torch.Tensor,
                          x_pos: torch.Tensor,
                          x_part: torch.Tensor,
                          x_neg: torch.Tensor,
                          gamma: float = DEFAULT_GAMMA,
                          margin_pos_neg: float = 1.0,
                          margin_pos_part: float = 0.5,
                          margin_part_neg: float = 0.5,
                          p: float = 2.0,
                          swap: bool = False,
                          reduction: str = "mean") -> torch.Tensor:
    """a + gamma*b + (1-gamma)*c with a,b,c = triplet margin terms (pos/neg, part/neg, pos/part)."""
    _require_gamma(gamma)
    _require_positive("margin_pos_neg", margin_pos_neg)
    _require_positive("margin_pos_part", margin_pos_part)
    _require_positive("margin_part_neg", margin_part_neg)
    _require_reduction(reduction)
    _require_positive("p", p)
    xs = (x_anchor, x_pos, x_part, x_neg)
    if any(x.dim() != 2 for x in xs) or any(x.shape != x_anchor.shape for x in xs):
        raise ValueError("x_anchor, x_pos, x_part and x_neg must all have the same shape (B, D)")
    if not all(x.is_cuda for x in xs):
        raise _lib.QstError("gamma_quadruplet_loss runs on the HIP device only (inputs are CPU tensors; no CPU path)")
    hp = (gamma, margin_pos_neg, margin_pos_part, margin_part_neg, p, swap)
    return _QuadrupletLossFn.apply(x_anchor, x_pos, x_part, x_neg, hp, _RED_CODE[reduction])


class QuadrupletLoss(torch.nn.Module, ABC):
    """Hyper-parameter holder with validated properties (reference: losses.py:157-238)."""

    def __init__(self,
                 margin_pos_neg: float = 1.0,
                 margin_pos_part: float = 1.0,
                 p: float = 2.0,
                 swap: bool = False,
                 reduction: str = "mean"):
        super().__init__()
        self._hp = {}
        self.margin_pos_neg = margin_pos_neg
        self.margin_pos_part = margin_pos_part
        self.reduction = reduction
        self.p = p
        self.swap = swap

    margin_pos_neg = property(lambda self: self._hp["margin_pos_neg"])
    margin_pos_part = property(lambda self: self._hp["margin_pos_part"])
    p = property(lambda self: self._hp["p"])
    swap = property(lambda self: self._hp["swap"])
    reduction = property(lambda self: self._hp["reduction"])

    @margin_pos_neg.setter
    def margin_pos_neg(self, v: float):
        _require_positive("margin_pos_neg", v)
        self._hp["margin_pos_neg"] = v

    @margin_pos_part.setter
    def margin_pos_part(self, v: float):
        _require_positive("margin_pos_part", v)
        self._hp["margin_pos_part"] = v

    @p.setter
    def p(self, v: float):
        _require_positive("p", v)
        self._hp["p"] = v

    @swap.setter
    def swap(self, v: bool):
        self._hp["swap"] = v

    @reduction.setter
    def reduction(self, v: str):
        _require_reduction(v)
        self._hp["reduction"] = v

    @abstractmethod
    def forward(self, x_anchor, x_pos, x_part, x_neg, reduction: Optional[str] = None, **kwargs) -> torch.Tensor:
        raise NotImplementedError()


class GammaQuadrupletLoss(QuadrupletLoss):
    """reference: losses.py:241-303 (class defaults: gamma 0.6, all three margins 1.0)."""

    def __init__(self,
                 gamma: float = DEFAULT_GAMMA,
                 margin_pos_neg: float = 1.0,
                 margin_pos_part: float = 1.0,
                 margin_part_neg: float = 1.0,
                 p: float = 2.0,
                 swap: bool = False,
                 reduction: str = "mean"):
        super().__init__(margin_pos_part=margin_pos_part, margin_pos_neg=margin_pos_neg, p=p, swap=swap,
                         reduction=reduction)
        self.gamma = gamma
        self.margin_part_neg = margin_part_neg

    gamma = property(lambda self: self._hp["gamma"])
    margin_part_neg = property(lambda self: self._hp["margin_part_neg"])

    @gamma.setter
    def gamma(self, v: float):
        _require_gamma(v)
        self._hp["gamma"] = v

    @margin_part_neg.setter
    def margin_part_neg(self, v: float):
        _require_positive("margin_part_neg", v)
        self._hp["margin_part_neg"] = v

    def forward(self, x_anchor, x_pos, x_part, x_neg, reduction: Optional[str] = None, **kwargs) -> torch.Tensor:
        return gamma_quadruplet_loss(x_anchor=x_anchor, x_pos=x_pos, x_part=x_part, x_neg=x_neg,
                                     gamma=self.gamma, margin_pos_neg=self.margin_pos_neg,
                                     margin_pos_part=self.margin_pos_part, margin_part_neg=self.margin_part_neg,
                                     p=self.p, swap=self.swap,
                                     reduction=self.reduction if reduction is None else reduction)
