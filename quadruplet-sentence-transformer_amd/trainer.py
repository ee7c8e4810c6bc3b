"""Fused quadruplet training step on one GPU per process, data-parallel over RCCL/xGMI.

One step = what SentenceTransformer.fit's inner loop does for this repo's objective
(/root/reference/training/main.py:128-148):

    loss_model(features, labels)   -> 4 encoder calls + GammaQuadrupletLoss   (here: ONE [4B, L] pass)
    loss.backward()                -> staged HIP backward
    clip_grad_norm_ ; AdamW.step ; zero_grad ; scheduler.step

The four columns of a quadruplet batch are concatenated into one [4B, L] encoder pass
(mathematically identical in eval mode). Quadruplets are independent, so
ranks shard the batch and exchange nothing but the gradient arena: one sum all-reduce per
finished layer, launched while lower layers are still in backward.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from .config import EncoderConfig, build_layout
from .encoder import HipEncoder, quadruplet_loss_raw

_REDUCTION = {"none": 0, "sum": 1, "mean": 2}


def gradient_buckets(cfg: EncoderConfig) -> List[Tuple[int, int]]:
    """Arena [begin, end) ranges in the order backward finishes them: layer N-1 ... layer 0, embeddings."""
    segs, total = build_layout(cfg)
    first = {}
    for s in segs:
        if s.name.startswith("layer."):
            l = int(s.name.split(".")[1])
            first.setdefault(l, s.offset)
    starts = [first[l] for l in range(cfg.num_layers)] + [total]
    out = [(starts[l], starts[l + 1]) for l in range(cfg.num_layers - 1, -1, -1)]
    out.append((0, starts[0]))
    return out


def allreduce_ranges(flat: torch.Tensor, ranges: Sequence[Tuple[int, int]], group=None, async_op: bool = False):
    """Sum all-reduce of slices of a flat tensor (device-agnostic: nccl/RCCL on GPU, gloo in CPU tests)."""
    import torch.distributed as dist
    works = []
    for b, e in ranges:
        w = dist.all_reduce(flat[b:e], op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            works.append(w)
    return works


def warmup_linear_lr(base_lr: float, step: int, warmup_steps: int, t_total: int) -> float:
    """transformers.get_linear_schedule_with_warmup, the 'WarmupLinear' scheduler ST fit() builds
    (SURVEY.md 3.1). `step` = number of scheduler steps already taken."""
    if step < warmup_steps:
        return base_lr * float(step) / float(max(1, warmup_steps))
    return base_lr * max(0.0, float(t_total - step) / float(max(1, t_total - warmup_steps)))


class QuadrupletTrainer:
    def __init__(self, cfg: EncoderConfig, arena=None, device=None, gamma: float = 0.6, margin_pos_neg: float = 1.0,
                 margin_pos_part: float = 0.5, margin_part_neg: float = 0.5, p: float = 2.0, swap: bool = False,
                 lr: float = 2e-5, weight_decay: float = 0.01, max_grad_norm: float = 1.0,
                 betas=(0.9, 0.999), eps: float = 1e-8, warmup_steps: int = 0, total_steps: int = 0,
                 process_group=None, world_size: int = 1, overlap: bool = True, encoder: Optional[HipEncoder] = None):
        self.cfg = cfg
        self.enc = encoder if encoder is not None else HipEncoder(cfg, device=device)
        if arena is not None:
            self.enc.load_arena(arena)
        self.enc.ensure_train_state()
        self.loss_args = (gamma, margin_pos_neg, margin_pos_part, margin_part_neg, p, swap)
        self.lr, self.wd, self.max_grad_norm, self.betas, self.eps = lr, weight_decay, max_grad_norm, betas, eps
        self.warmup_steps, self.total_steps = warmup_steps, total_steps
        self.sched_step = 0
        self.group, self.world = process_group, world_size
        self.overlap = overlap
        self.buckets = gradient_buckets(cfg)

    def current_lr(self) -> float:
        if self.total_steps <= 0:
            return self.lr
        return warmup_linear_lr(self.lr, self.sched_step, self.warmup_steps, self.total_steps)

    def forward_loss(self, ids4: torch.Tensor, mask4: torch.Tensor, types4: Optional[torch.Tensor] = None,
                     training: bool = False, want_grads: bool = False):
        """ids4/mask4 int64 [4, B, L] on the encoder's device. Returns (loss [1], emb [4,B,H], grads, saved, flat inputs)."""
        four, B, L = ids4.shape
        assert four == 4
        ids = ids4.reshape(4 * B, L)
        mask = mask4.reshape(4 * B, L)
        types = types4.reshape(4 * B, L) if (types4 is not None and self.cfg.type_vocab_size > 0) else None
        emb, _, saved = self.enc.forward(ids, mask, types, training=training)
        e4 = emb.view(4, B, -1)
        loss, g = quadruplet_loss_raw(e4[0], e4[1], e4[2], e4[3], *self.loss_args, _REDUCTION["mean"],
                                      want_grads=want_grads)
        return loss, e4, g, saved, (ids, mask, types)

    def step(self, ids4: torch.Tensor, mask4: torch.Tensor, types4: Optional[torch.Tensor] = None) -> torch.Tensor:
        enc = self.enc
        loss, _, g, saved, (ids, mask, types) = self.forward_loss(ids4, mask4, types4, training=True, want_grads=True)
        grad_emb = torch.cat(g, 0)
        n, L = ids.shape
        lib, st = enc.lib, _lib.current_stream_ptr()
        ws = enc._arena("_ws", lib.qst_encoder_bwd_workspace_bytes(enc.handle, n, L))
        N = self.cfg.num_layers

        def stage(head, hi, lo, emb_):
            _lib.check(lib.qst_encoder_backward_partial(
                enc.handle, ids.data_ptr(), mask.data_ptr(), _lib.ptr(types), n, L, enc.params.data_ptr(),
                enc.shadow.data_ptr(), grad_emb.data_ptr(), enc.grads.data_ptr(), saved.data_ptr(), saved.numel(),
                ws.data_ptr(), ws.numel(), int(head), hi, lo, int(emb_), st), "qst_encoder_backward_partial")

        if self.world > 1 and self.overlap:
            works = []
            for k, l in enumerate(range(N - 1, -1, -1)):
                stage(k == 0, l + 1, l, False)
                works += allreduce_ranges(enc.grads, [self.buckets[k]], self.group, async_op=True)
            stage(False, 0, 0, True)
            works += allreduce_ranges(enc.grads, [self.buckets[N]], self.group, async_op=True)
            for w in works:
                w.wait()
        else:
            stage(True, N, 0, True)
            if self.world > 1:
                allreduce_ranges(enc.grads, [(0, enc.total)], self.group)
        enc.adamw_step(self.current_lr(), self.betas, self.eps, self.wd, self.max_grad_norm, 1.0 / self.world)
        self.sched_step += 1
        return loss
