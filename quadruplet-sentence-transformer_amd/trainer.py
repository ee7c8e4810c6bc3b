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
from .encoder import HipEncoder, quadruplet_loss_raw, stacked

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
    """Sum all-reduce of slices of a flat tensor (device-agnostic: nccl/RCCL on GPU, gloo in CPU tests). `group`: a
    torch.distributed process group (None = the default one) or a comm.NativeComm (libqst.so's own RCCL binding)."""
    works = []
    if hasattr(group, "all_reduce"):                      # comm.NativeComm
        for b, e in ranges:
            w = group.all_reduce(flat[b:e], async_op=async_op)
            if async_op:
                works.append(w)
        return works
    import torch.distributed as dist
    for b, e in ranges:
        w = dist.all_reduce(flat[b:e], op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            works.append(w)
    return works


BWD_HEAD, BWD_EMBED, BWD_SKIP_WGRAD, BWD_WGRAD_ONLY = 1, 2, 4, 8      # include/qst.h: QST_BWD_*


def staged_reduce_order(buckets: Sequence[Tuple[int, int]], total: int, overlap: bool) -> List[Tuple[int, int]]:
    """The arena ranges staged_backward all-reduces, in the order it issues them. A rank that has nothing to back-propagate
    in a step (its shard of the last batch is empty) must still take part in exactly this sequence of collectives: a
    different number or size of all-reduces than its peers' hangs RCCL (gloo raises a size error)."""
    if not overlap:
        return [(0, total)]
    N = len(buckets) - 1
    return [buckets[k] for k in range(N - 1)] + [buckets[N], buckets[N - 1]]


def staged_backward(enc: HipEncoder, ids, mask, types, grad_emb, saved, ws=None, buckets=None, group=None,
                    overlap: bool = True, precision: str = "bf16"):
    """One backward pass of the encoder with the data-parallel gradient exchange inside it.

    Stages run top-down on the compute stream; after each finished layer its contiguous slice of the gradient arena
    goes into an asynchronous sum all-reduce (torch.distributed: RCCL over xGMI on its own stream), so the exchange of
    layer l overlaps the backward of layers l-1 ... 0. The tail is ordered for the largest bucket: layer 0 runs WITHOUT
    its weight-gradient launch, the embedding stage follows, the embedding bucket (word table = half of a MiniLM arena)
    starts its all-reduce, and only then does layer 0's grouped weight-gradient kernel run -- underneath that exchange --
    followed by layer 0's own (small) bucket. Returns the list of pending works; the caller waits on them before the
    optimiser step (the global-norm clip needs every reduced gradient, so replicas stay bit-identical).

    buckets=None (single process): one call, no exchange.
    precision: "bf16"; "f16" (the same stages on IEEE-half operands: `saved` from forward(training=True, precision="f16"),
    grad_emb carrying the loss scale); or "fp8" when `saved` was filled by forward(training=True, precision="fp8") -- the
    bf16 stages on that handle. What a training forward did with dropout is recorded per activation arena, process-wide, so either
    handle rebuilds the masks of the forward that filled `saved` (an arena no training forward has filled is refused)."""
    lib, st = enc.lib, _lib.current_stream_ptr()
    n, L = ids.shape
    if precision not in ("bf16", "fp8", "f16", "f16w", "bf16x3"):
        raise ValueError(f"unknown precision {precision!r}")
    handle = enc._handle_for(precision)
    x3 = precision == "bf16x3"
    if ws is None:
        ws = enc._arena("_ws_x3" if x3 else "_ws", lib.qst_encoder_bwd_workspace_bytes(handle, n, L))
    grad_emb = grad_emb.contiguous()
    N = enc.cfg.num_layers

    def stage(flags, hi, lo):
        _lib.check(lib.qst_encoder_backward_stage(
            handle, ids.data_ptr(), mask.data_ptr(), _lib.ptr(types), n, L, enc.params.data_ptr(),
            enc.shadow_for(handle).data_ptr(), grad_emb.data_ptr(), enc.grads.data_ptr(), saved.data_ptr(), saved.numel(),
            ws.data_ptr(), ws.numel(), int(flags), hi, lo, st), "qst_encoder_backward_stage")

    if buckets is None:
        stage(BWD_HEAD | BWD_EMBED, N, 0)
        return []
    order = staged_reduce_order(buckets, enc.total, overlap)          # (the one place that fixes the collective sequence)
    if not overlap:
        stage(BWD_HEAD | BWD_EMBED, N, 0)
        allreduce_ranges(enc.grads, order, group)
        return []
    works = []
    for k, l in enumerate(range(N - 1, 0, -1)):                        # layers N-1 ... 1
        stage(BWD_HEAD if k == 0 else 0, l + 1, l)
        works += allreduce_ranges(enc.grads, [order[k]], group, async_op=True)
    if x3:
        # the parity path has no postponed weight-gradient launch: layer 0 and the embeddings in one stage, then the same two
        # buckets in the same order (round 5: rounds 3-4 reduced everything after a one-call backward)
        stage((BWD_HEAD if N == 1 else 0) | BWD_EMBED, 1, 0)
        works += allreduce_ranges(enc.grads, [order[N - 1]], group, async_op=True)
        works += allreduce_ranges(enc.grads, [order[N]], group, async_op=True)
        return works
    stage((BWD_HEAD if N == 1 else 0) | BWD_SKIP_WGRAD | BWD_EMBED, 1, 0)      # layer 0 dgrads + embeddings
    works += allreduce_ranges(enc.grads, [order[N - 1]], group, async_op=True)   # embedding bucket: the big one
    stage(BWD_WGRAD_ONLY, 1, 0)                                               # layer 0 weight gradients, under it
    works += allreduce_ranges(enc.grads, [order[N]], group, async_op=True)
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
                 process_group=None, world_size: int = 1, overlap: bool = True, encoder: Optional[HipEncoder] = None,
                 use_graph: bool = False, dropout=None, dropout_seed: int = 0, force_dp: bool = False,
                 precision: str = "bf16", amp_init_scale: float = 65536.0, amp_growth_interval: int = 2000):
        """precision: "bf16" (the throughput path); "f16" -- the same kernels on IEEE-half operands (11 significand bits:
        embeddings inside the north-star tolerance for the six-layer models) under a dynamic loss scale on the device, i.e.
        what the reference's `use_amp=True` is (autocast + GradScaler, training/main.py:142): amp_init_scale and
        amp_growth_interval are GradScaler's init_scale / growth_interval (<= 0: a static scale); "f16w" -- f16 whose forward
        Linears multiply by hi + lo of every weight (split-f16 weights: the weight rounding, which does not average out over
        the tokens of a sequence, is gone; a second pass over K per forward GEMM); "fp8" -- BASELINE configs[4]: the forward's Linears on the fp8 matrix
        cores (MXFP8 weights and activations), dgrad / wgrad in bf16 from the fp32 master weights (H and I multiples of 128;
        dropout as on the bf16 path); or "bf16x3" -- the parity path: fp32 activations, every product as three
        split-bf16 MFMAs, gradients fp32-class (the reference trains in fp32, training/main.py:142). Several times slower;
        data-parallel too: its backward runs in the same stages as the bf16 one (round 5), a finished layer's bucket is
        all-reduced under the layers below.
        dropout: None / 0 = off; a float p = HF's hidden_dropout_prob = attention_probs_dropout_prob = p; a pair
        (p_hidden, p_attn). The reference's fit() trains with 0.1 (HF config defaults, train() mode). Ranks of a
        data-parallel job should pass different dropout_seed values (fit() adds the rank)."""
        self.cfg = cfg
        if precision not in ("bf16", "f16", "f16w", "bf16x3", "fp8"):
            raise ValueError("training precision is 'bf16', 'f16', 'f16w', 'bf16x3' or 'fp8'")
        if precision == "bf16x3" and use_graph:
            raise ValueError("precision='bf16x3' (the parity path) trains without a graph")
        if precision == "fp8" and use_graph:
            raise ValueError("precision='fp8' (fp8 forward GEMMs, bf16 backward) trains without a graph")
        self.precision = precision
        self.enc = encoder if encoder is not None else HipEncoder(cfg, device=device)
        if arena is not None:
            self.enc.load_arena(arena)
        self.enc.ensure_train_state()
        self.amp_growth_interval = int(amp_growth_interval)
        self.amp = precision in ("f16", "f16w")      # IEEE-half operands: the step runs under the device-side loss scaler
        if self.amp:
            self.enc.ensure_amp_scaler(amp_init_scale)
        if dropout is not None:
            ph, pa = (dropout, dropout) if isinstance(dropout, (int, float)) else dropout
            self.enc.set_dropout(float(ph), float(pa), int(dropout_seed))
        self.loss_args = (gamma, margin_pos_neg, margin_pos_part, margin_part_neg, p, swap)
        self.lr, self.wd, self.max_grad_norm, self.betas, self.eps = lr, weight_decay, max_grad_norm, betas, eps
        self.warmup_steps, self.total_steps = warmup_steps, total_steps
        self.sched_step = 0
        self.group, self.world = process_group, world_size
        # force_dp: run the data-parallel step (staged backward + one asynchronous all-reduce per bucket) even with one rank --
        # the RCCL code path on a box with a single GPU (bench.py "dp_rccl_ws1", tests)
        self.force_dp = bool(force_dp)
        self.overlap = overlap
        self.buckets = gradient_buckets(cfg)
        # use_graph: single-GPU steps are captured once per (B, L) into a HIP graph (torch.cuda.CUDAGraph over the
        # stream the C-ABI launches on) and replayed -- ~100 launches per step become one; the schedule and the step
        # counter live on the device (qst_clip_adamw_step_sched). Small batches are launch-bound without it
        # (MiniLM B=8 L=32: 1.9 ms/step eager).
        if use_graph and world_size > 1:
            raise ValueError("use_graph covers the single-GPU step; the DP step interleaves all-reduces with backward")
        self.use_graph = use_graph
        self._graphs = {}

    def current_lr(self) -> float:
        if self.total_steps <= 0:
            return self.lr
        return warmup_linear_lr(self.lr, self.sched_step, self.warmup_steps, self.total_steps)

    def forward_loss(self, ids4: torch.Tensor, mask4: torch.Tensor, types4: Optional[torch.Tensor] = None,
                     training: bool = False, want_grads: bool = False, saved: Optional[torch.Tensor] = None,
                     precision: str = "bf16"):
        """ids4/mask4 int64 [4, B, L] on the encoder's device. Returns (loss [1], emb [4,B,H], grads, saved, flat inputs)."""
        four, B, L = ids4.shape
        assert four == 4
        ids = ids4.reshape(4 * B, L)
        mask = mask4.reshape(4 * B, L)
        types = types4.reshape(4 * B, L) if (types4 is not None and self.cfg.type_vocab_size > 0) else None
        emb, _, saved = self.enc.forward(ids, mask, types, training=training, saved=saved, precision=precision)
        e4 = emb.view(4, B, -1)
        # f16 training: the loss gradient leaves the loss kernel multiplied by the device-resident loss scale (grad_out = the
        # scaler's first word), so every f16 gradient tensor of the backward sits inside half's range
        gout = self.enc.amp_scaler if (want_grads and precision in ("f16", "f16w")) else None
        loss, g = quadruplet_loss_raw(e4[0], e4[1], e4[2], e4[3], *self.loss_args, _REDUCTION["mean"],
                                      grad_out=gout, want_grads=want_grads)
        return loss, e4, g, saved, (ids, mask, types)

    def step(self, ids4: torch.Tensor, mask4: torch.Tensor, types4: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.use_graph:
            return self._step_graph(ids4, mask4, types4)
        return self._step_eager(ids4, mask4, types4, sched_on_device=False)

    def _step_graph(self, ids4, mask4, types4):
        key = (tuple(ids4.shape), types4 is not None)
        ent = self._graphs.get(key)
        if ent is None:
            # first batch of this shape: run it eagerly (sizes every arena, sets kernel attributes), then capture
            static = [ids4.clone(), mask4.clone(), None if types4 is None else types4.clone()]
            # the graph owns its activation arena and workspace: the encoder's shared ones may be re-allocated by a
            # later, larger call (encode() uses them too), which would leave a captured graph with dangling pointers
            n, L = 4 * ids4.shape[1], ids4.shape[2]
            enc = self.enc
            bufs = dict(
                saved=torch.empty(enc.lib.qst_encoder_saved_bytes(enc._handle_for(self.precision), n, L, 1), dtype=torch.uint8, device=enc.device),
                ws=torch.empty(enc.lib.qst_encoder_bwd_workspace_bytes(enc._handle_for(self.precision), n, L), dtype=torch.uint8, device=enc.device))
            loss = self._step_eager(*static, sched_on_device=True, **bufs)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_loss = self._step_eager(*static, sched_on_device=True, count=False, **bufs)
            self._graphs[key] = (graph, static, static_loss, bufs)
            return loss
        graph, static, static_loss, _ = ent
        static[0].copy_(ids4, non_blocking=True)
        static[1].copy_(mask4, non_blocking=True)
        if types4 is not None:
            static[2].copy_(types4, non_blocking=True)
        graph.replay()
        self.enc.opt_step += 1
        # the captured step ends with the refresh of the operand shadow it trains on (the other one is stale)
        self.enc.shadow_stale = self.precision != "bf16"
        self.enc.shadow_f16_stale = self.precision != "f16"
        self.enc.shadow_f16w_stale = self.precision != "f16w"
        self.sched_step += 1
        return static_loss

    def _step_eager(self, ids4: torch.Tensor, mask4: torch.Tensor, types4: Optional[torch.Tensor] = None,
                    sched_on_device: bool = False, count: bool = True, saved: Optional[torch.Tensor] = None,
                    ws: Optional[torch.Tensor] = None) -> torch.Tensor:
        enc = self.enc
        loss, _, g, saved, (ids, mask, types) = self.forward_loss(ids4, mask4, types4, training=True, want_grads=True,
                                                                 saved=saved, precision=self.precision)
        if self.precision == "bf16x3" and not (self.world > 1 or self.force_dp):
            enc.backward(ids, mask, types, stacked(g), saved, precision="bf16x3")
            works = []
        elif self.precision == "fp8" and not (self.world > 1 or self.force_dp):
            enc.backward(ids, mask, types, stacked(g), saved, precision="fp8")
            works = []
        else:
            works = staged_backward(enc, ids, mask, types, stacked(g), saved, ws,
                                    self.buckets if (self.world > 1 or self.force_dp) else None, self.group, self.overlap,
                                    precision=self.precision)
        for w in works:
            w.wait()
        if self.amp:
            opt0 = enc.opt_step
            enc.adamw_step_amp(self.lr, self.warmup_steps, self.total_steps, self.betas, self.eps, self.wd,
                               self.max_grad_norm, 1.0 / self.world, growth_interval=self.amp_growth_interval)
            if sched_on_device:
                # a graph replays the whole step: the next forward's operand refresh belongs inside it
                enc.refresh_shadow_f16() if self.precision == "f16" else enc.refresh_shadow_f16w()
            if not count:
                enc.opt_step = opt0
        elif sched_on_device:
            opt0 = enc.opt_step
            enc.adamw_step_sched(self.lr, self.warmup_steps, self.total_steps, self.betas, self.eps, self.wd,
                                 self.max_grad_norm, 1.0 / self.world)
            # a graph replays the whole step, so the next forward's shadow refresh belongs inside it
            enc.refresh_shadow()
            if not count:
                enc.opt_step = opt0          # capture pass: nothing executed
        else:
            enc.adamw_step(self.current_lr(), self.betas, self.eps, self.wd, self.max_grad_norm, 1.0 / self.world)
        if count:
            self.sched_step += 1
        return loss
