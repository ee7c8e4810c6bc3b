"""HipEncoder: torch tensors as storage, libqst.so as the only executor.

Owns the flat fp32 parameter / gradient / Adam-moment arenas and the bf16
operand shadow on one GPU and drives the C-ABI (include/qst.h). No
torch.nn.Module executes anything here; torch is used for device memory and
streams only (ownership rule).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .config import EncoderConfig, build_layout, hf_param_views


def _round_up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


class HipEncoder:
    def __init__(self, cfg: EncoderConfig, device: Optional[torch.device] = None, precision: int = 0):
        self.cfg = cfg
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.QstError("HipEncoder needs a HIP device; there is no CPU fallback")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        torch.cuda.set_device(self.device)
        self.ccfg = _lib.make_config(cfg, precision)
        self.segments, self.total = build_layout(cfg)
        n = self.lib.qst_arena_elems(self.ccfg)
        if n != self.total:
            raise _lib.QstError(f"arena layout mismatch: python {self.total} vs libqst {n}")
        h = _lib.vp()
        _lib.check(self.lib.qst_encoder_create(self.ccfg, h), "qst_encoder_create")
        self.handle = h
        self.dropout = None           # (p_hidden, p_attn, seed) once set_dropout() switched it on
        self.dropout_step = 0
        self.drop_state = None
        self.handle_x3 = None         # QST_PREC_BF16X3 handle over the SAME arenas, created on first use
        self.handle_mx = None         # QST_PREC_FP8 handle (MXFP8 weights and activations on the fp8 matrix cores, inference)
        self.handle_f16 = None        # QST_PREC_F16 handle: the bf16 kernels compiled on IEEE half (round 5), its own shadow
        self.shadow_f16: Optional[torch.Tensor] = None
        self.shadow_f16_stale = True
        self.handle_f16w = None       # QST_PREC_F16W: f16 with split weights (hi + lo) in the forward; shadow = [f16 arena | low halves]
        self.shadow_f16w: Optional[torch.Tensor] = None
        self.shadow_f16w_stale = True
        self.amp_scaler: Optional[torch.Tensor] = None   # device fp32 [4] {loss scale, growth tracker, last skipped, #skipped}
        self._step2_dev: Optional[torch.Tensor] = None   # device int64 [2] {optimiser steps, scheduler steps} of the amp step
        self.shadow_mx: Optional[torch.Tensor] = None
        self.shadow_mx_stale = True
        self.params = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.grads: Optional[torch.Tensor] = None
        self.exp_avg: Optional[torch.Tensor] = None
        self.exp_avg_sq: Optional[torch.Tensor] = None
        self.shadow = torch.zeros(self.lib.qst_shadow_elems(self.ccfg), dtype=torch.bfloat16, device=self.device)
        self.shadow_stale = True
        self._saved: Optional[torch.Tensor] = None
        self._ws: Optional[torch.Tensor] = None
        self._saved_x3: Optional[torch.Tensor] = None      # activation arena / scratch of the bf16x3 training path
        self._ws_x3: Optional[torch.Tensor] = None
        self._scratch = torch.zeros(2048, dtype=torch.float32, device=self.device)
        self._step_dev: Optional[torch.Tensor] = None    # device-side optimiser step counter (graph-captured steps)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.opt_step = 0

    def __del__(self):
        try:
            for attr in ("handle", "handle_x3", "handle_mx", "handle_f16", "handle_f16w"):
                if getattr(self, attr, None):
                    self.lib.qst_encoder_destroy(getattr(self, attr))
                    setattr(self, attr, None)
        except Exception:
            pass

    # ------------------------------------------------------------------ parameters
    def load_arena(self, arena) -> None:
        t = torch.as_tensor(np.asarray(arena), dtype=torch.float32) if not torch.is_tensor(arena) else arena
        if t.numel() != self.total:
            raise ValueError(f"arena has {t.numel()} elements, expected {self.total}")
        self.params.copy_(t.to(self.device))
        self.shadow_stale = True
        self.shadow_mx_stale = True
        self.shadow_f16_stale = True
        self.shadow_f16w_stale = True

    def named_views(self) -> Dict[str, torch.Tensor]:
        """HF-named views into the parameter arena (no copies)."""
        seg = {s.name: s for s in self.segments}
        out = {}
        for name, sname, off, shape in hf_param_views(self.cfg):
            s = seg[sname]
            n = int(np.prod(shape))
            out[name] = self.params[s.offset + off: s.offset + off + n].view(*shape)
        return out

    def grad_views(self) -> Dict[str, torch.Tensor]:
        self.ensure_train_state()
        seg = {s.name: s for s in self.segments}
        out = {}
        for name, sname, off, shape in hf_param_views(self.cfg):
            s = seg[sname]
            n = int(np.prod(shape))
            out[name] = self.grads[s.offset + off: s.offset + off + n].view(*shape)
        return out

    def ensure_train_state(self) -> None:
        if self.grads is None:
            self.grads = torch.zeros_like(self.params)
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(self.params)
            self.exp_avg_sq = torch.zeros_like(self.params)

    # ------------------------------------------------------------------ dropout
    def set_dropout(self, p_hidden: float = 0.0, p_attn: float = 0.0, seed: int = 0) -> None:
        """Dropout for training forwards/backwards (HF hidden_dropout_prob / attention_probs_dropout_prob; the reference's
        fit() trains in train() mode with 0.1 / 0.1). Masks are counter-based -- a function of (seed, step, tensor, element)
        recomputed by the backward kernels, never stored; `dropout_step` counts the training forwards run since this call
        (the device-side counter the kernels read). 0 / 0 switches dropout off. Inference forwards never drop."""
        if not (0.0 <= p_hidden < 1.0 and 0.0 <= p_attn < 1.0):
            raise ValueError("dropout probabilities must be in [0, 1)")
        on = p_hidden > 0.0 or p_attn > 0.0
        if on:
            if getattr(self, "drop_state", None) is None:
                self.drop_state = torch.zeros(4, dtype=torch.int32, device=self.device)
            _lib.check(self.lib.qst_dropout_init(self.drop_state.data_ptr(), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                                 _lib.current_stream_ptr()), "qst_dropout_init")
        for h in (self.handle, self.handle_mx, self.handle_x3, self.handle_f16, self.handle_f16w):      # every precision's training forward drops at the same places
            if h is not None:
                _lib.check(self.lib.qst_encoder_set_dropout(h, float(p_hidden), float(p_attn),
                                                            self.drop_state.data_ptr() if on else None), "qst_encoder_set_dropout")
        self.dropout = (float(p_hidden), float(p_attn), int(seed)) if on else None
        self.dropout_step = 0

    def set_ffn_chain(self, mask: int) -> None:
        """Where the feed-forward block runs as one kernel (include/qst.h qst_encoder_set_ffn_chain): bit 0 inference
        forward (default), bit 1 training forward, bit 2 backward."""
        _lib.check(self.lib.qst_encoder_set_ffn_chain(self.handle, int(mask)), "qst_encoder_set_ffn_chain")

    def set_ln_fusion(self, mode: int) -> None:
        """Where a projection + LayerNorm (and a dgrad + LayerNorm backward) run as one kernel (include/qst.h
        qst_encoder_set_ln_fusion): 0 by size (default), 1 wherever such a kernel exists, 2 never. Every precision's handle."""
        for h in (self.handle, self.handle_mx, self.handle_x3, self.handle_f16, self.handle_f16w):
            if h is not None:
                _lib.check(self.lib.qst_encoder_set_ln_fusion(h, int(mode)), "qst_encoder_set_ln_fusion")
        self.ln_fusion = int(mode)

    def set_dropout_step(self, step: int) -> None:
        """Continue the mask stream at `step` training forwards (checkpoint resume)."""
        if self.dropout is not None:
            self.drop_state[2] = int(step)
            self.dropout_step = int(step)

    def refresh_shadow_mx(self) -> None:
        """Quantise every Linear weight to MXFP8 (e4m3 + one E8M0 scale per 32 input features; QST_PREC_FP8)."""
        _lib.check(self.lib.qst_refresh_shadow_mx(self.handle_mx, self.params.data_ptr(), self.shadow_mx.data_ptr(),
                                                  _lib.current_stream_ptr()), "qst_refresh_shadow_mx")
        self.shadow_mx_stale = False

    def refresh_shadow_f16(self) -> None:
        """[W | W^T] of every Linear weight as IEEE half (QST_PREC_F16: qst_refresh_shadow on that handle)."""
        _lib.check(self.lib.qst_refresh_shadow(self.handle_f16, self.params.data_ptr(), self.shadow_f16.data_ptr(),
                                               _lib.current_stream_ptr()), "qst_refresh_shadow(f16)")
        self.shadow_f16_stale = False

    def refresh_shadow_f16w(self) -> None:
        """QST_PREC_F16W: the f16 [W | W^T] arena and, behind it, the low halves of the split weights."""
        _lib.check(self.lib.qst_refresh_shadow(self.handle_f16w, self.params.data_ptr(), self.shadow_f16w.data_ptr(),
                                               _lib.current_stream_ptr()), "qst_refresh_shadow(f16w)")
        self.shadow_f16w_stale = False

    def refresh_shadow(self) -> None:
        _lib.check(self.lib.qst_refresh_shadow(self.handle, self.params.data_ptr(), self.shadow.data_ptr(),
                                               _lib.current_stream_ptr()), "qst_refresh_shadow")
        self.shadow_stale = False

    # ------------------------------------------------------------------ shapes
    @staticmethod
    def pad_inputs(ids: torch.Tensor, mask: torch.Tensor, type_ids: Optional[torch.Tensor], pad_id: int
                   ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor], int]:
        """Pad L up to a multiple of 32 with masked positions (outputs do not depend on padded content)."""
        n, L = ids.shape
        Lp = _round_up(L, 32)
        if Lp != L:
            ids = torch.nn.functional.pad(ids, (0, Lp - L), value=pad_id)
            mask = torch.nn.functional.pad(mask, (0, Lp - L), value=0)
            if type_ids is not None:
                type_ids = torch.nn.functional.pad(type_ids, (0, Lp - L), value=0)
        return ids.contiguous(), mask.contiguous(), None if type_ids is None else type_ids.contiguous(), L

    def _arena(self, attr: str, nbytes: int) -> torch.Tensor:
        buf = getattr(self, attr)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            setattr(self, attr, buf)
        return buf

    # ------------------------------------------------------------------ forward / backward
    def _handle_for(self, precision: str):
        h = self._handle_for_raw(precision)
        if getattr(self, "ln_fusion", 0) and not getattr(h, "_ln_fusion_set", None) == self.ln_fusion:
            _lib.check(self.lib.qst_encoder_set_ln_fusion(h, self.ln_fusion), "qst_encoder_set_ln_fusion")
            try:
                h._ln_fusion_set = self.ln_fusion
            except AttributeError:
                pass
        return h

    def _handle_for_raw(self, precision: str):
        if precision in ("bf16", 0, None):
            return self.handle
        if precision in ("fp8", 3):
            if self.handle_mx is None:
                h = _lib.vp()
                _lib.check(self.lib.qst_encoder_create(_lib.make_config(self.cfg, 3), h), "qst_encoder_create(fp8)")
                self.handle_mx = h
                self.shadow_mx = torch.zeros(self.lib.qst_shadow8_bytes(self.ccfg), dtype=torch.uint8, device=self.device)
                self.shadow_mx_stale = True
                if self.dropout is not None:         # (created after set_dropout: same settings, same device counter)
                    _lib.check(self.lib.qst_encoder_set_dropout(h, self.dropout[0], self.dropout[1], self.drop_state.data_ptr()),
                               "qst_encoder_set_dropout")
            return self.handle_mx
        if precision in ("f16", "fp16", 4):
            if self.handle_f16 is None:
                h = _lib.vp()
                _lib.check(self.lib.qst_encoder_create(_lib.make_config(self.cfg, 4), h), "qst_encoder_create(f16)")
                self.handle_f16 = h
                self.shadow_f16 = torch.zeros(self.lib.qst_shadow_elems(self.ccfg), dtype=torch.float16, device=self.device)
                self.shadow_f16_stale = True
                if self.dropout is not None:
                    _lib.check(self.lib.qst_encoder_set_dropout(h, self.dropout[0], self.dropout[1], self.drop_state.data_ptr()),
                               "qst_encoder_set_dropout")
            return self.handle_f16
        if precision in ("f16w", 5):
            if self.handle_f16w is None:
                h = _lib.vp()
                c5 = _lib.make_config(self.cfg, 5)
                _lib.check(self.lib.qst_encoder_create(c5, h), "qst_encoder_create(f16w)")
                self.handle_f16w = h
                self.shadow_f16w = torch.zeros(self.lib.qst_shadow_elems(c5), dtype=torch.float16, device=self.device)
                self.shadow_f16w_stale = True
                if self.dropout is not None:
                    _lib.check(self.lib.qst_encoder_set_dropout(h, self.dropout[0], self.dropout[1], self.drop_state.data_ptr()),
                               "qst_encoder_set_dropout")
            return self.handle_f16w
        if precision not in ("bf16x3", 1):
            raise ValueError(f"unknown precision {precision!r} (bf16 | f16 | f16w | bf16x3 | fp8)")
        if self.handle_x3 is None:
            h = _lib.vp()
            _lib.check(self.lib.qst_encoder_create(_lib.make_config(self.cfg, 1), h), "qst_encoder_create(x3)")
            self.handle_x3 = h
            if self.dropout is not None:
                _lib.check(self.lib.qst_encoder_set_dropout(h, self.dropout[0], self.dropout[1], self.drop_state.data_ptr()),
                           "qst_encoder_set_dropout")
        return self.handle_x3

    def forward(self, ids: torch.Tensor, mask: torch.Tensor, type_ids: Optional[torch.Tensor] = None,
                training: bool = False, want_tokens: bool = False, saved: Optional[torch.Tensor] = None,
                precision: str = "bf16"):
        """ids/mask int64 [n, L] on this device, L % 32 == 0. Returns (emb [n,H], tok [n,L,H] or None, saved).
        precision="bf16x3" runs the fp32-class parity path, "fp8" the fp8 matrix-core path (MXFP8 weights and
        activations). training=True keeps what the matching backward(precision=...) needs."""
        assert ids.dtype == torch.int64 and mask.dtype == torch.int64 and ids.is_cuda and ids.is_contiguous()
        n, L = ids.shape
        handle = self._handle_for(precision)
        if self.shadow_stale and (handle is self.handle or (training and handle is self.handle_mx)):
            self.refresh_shadow()                    # (an fp8 training forward: its backward runs on the bf16 shadows)
        shadow = self.shadow
        if handle is self.handle_mx and handle is not None:
            if self.shadow_mx_stale:
                self.refresh_shadow_mx()
            shadow = self.shadow_mx
        if handle is self.handle_f16 and handle is not None:
            if self.shadow_f16_stale:
                self.refresh_shadow_f16()
            shadow = self.shadow_f16
        if handle is self.handle_f16w and handle is not None:
            if self.shadow_f16w_stale:
                self.refresh_shadow_f16w()
            shadow = self.shadow_f16w
        nbytes = self.lib.qst_encoder_saved_bytes(handle, n, L, int(training))
        if nbytes == 0:
            raise _lib.QstError(f"unsupported shape nseq={n} L={L} for this encoder (L % 32 == 0, L <= 512)")
        if saved is None:
            saved = self._arena("_saved_x3" if handle in (self.handle_x3, self.handle_mx) and training else "_saved", nbytes)
        emb = torch.empty(n, self.cfg.hidden_size, dtype=torch.float32, device=self.device)
        tok = torch.empty(n, L, self.cfg.hidden_size, dtype=torch.float32, device=self.device) if want_tokens else None
        _lib.check(self.lib.qst_encoder_forward(
            handle, ids.data_ptr(), mask.data_ptr(), _lib.ptr(type_ids), n, L, self.params.data_ptr(),
            shadow.data_ptr(), emb.data_ptr(), _lib.ptr(tok), saved.data_ptr(), saved.numel(), int(training),
            _lib.current_stream_ptr()), "qst_encoder_forward")
        if training and self.dropout is not None:
            self.dropout_step += 1           # mirrors the device counter (tests rebuild this step's masks from it)
        return emb, tok, saved

    def shadow_for(self, handle) -> torch.Tensor:
        """The [W | W^T] operand shadow a BACKWARD on `handle` reads: IEEE half for the f16 handle, bf16 otherwise."""
        if handle is not None and handle is self.handle_f16w:
            return self.shadow_f16w
        return self.shadow_f16 if (handle is self.handle_f16 and handle is not None) else self.shadow

    def backward(self, ids, mask, type_ids, grad_emb: torch.Tensor, saved: torch.Tensor, precision: str = "bf16") -> None:
        """Accumulate d(loss)/d(params) into self.grads given d(loss)/d(emb). precision="bf16x3": the fp32-class backward
        of a forward(training=True, precision="bf16x3") (the parity path: one call); "fp8": the bf16
        backward over what a forward(training=True, precision="fp8") kept (fp8 forward GEMMs, bf16 dgrad / wgrad)."""
        self.ensure_train_state()
        n, L = ids.shape
        handle = self._handle_for(precision)
        if handle is self.handle_mx and self.shadow_stale:
            self.refresh_shadow()                    # the backward's operands are the bf16 shadows
        nws = self.lib.qst_encoder_bwd_workspace_bytes(handle, n, L)
        ws = self._arena("_ws_x3" if handle is self.handle_x3 else "_ws", nws)
        grad_emb = grad_emb.contiguous()
        _lib.check(self.lib.qst_encoder_backward(
            handle, ids.data_ptr(), mask.data_ptr(), _lib.ptr(type_ids), n, L, self.params.data_ptr(),
            self.shadow_for(handle).data_ptr(), grad_emb.data_ptr(), self.grads.data_ptr(), saved.data_ptr(), saved.numel(),
            ws.data_ptr(), ws.numel(), _lib.current_stream_ptr()), "qst_encoder_backward")

    def adamw_step(self, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.01,
                   max_grad_norm: float = 1.0, grad_scale: float = 1.0) -> None:
        """clip_grad_norm_ + AdamW + zero_grad in one pass over the arena; the norm stays on the device."""
        self.ensure_train_state()
        self.opt_step += 1
        _lib.check(self.lib.qst_clip_adamw_step(
            self.handle, self.params.data_ptr(), self.grads.data_ptr(), self.exp_avg.data_ptr(),
            self.exp_avg_sq.data_ptr(), lr, betas[0], betas[1], eps, weight_decay, max_grad_norm, grad_scale,
            self.opt_step, self.grad_norm.data_ptr(), self._scratch.data_ptr(), _lib.current_stream_ptr()),
            "qst_clip_adamw_step")
        self.shadow_stale = True
        self.shadow_mx_stale = True
        self.shadow_f16_stale = True
        self.shadow_f16w_stale = True


    # ------------------------------------------------------------------ optimiser state (true resume)
    def optimizer_state(self) -> Dict[str, torch.Tensor]:
        """The Adam moments as flat arenas (same layout as the parameters) + the step counter."""
        self.ensure_train_state()
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "opt_step": torch.tensor([self.opt_step], dtype=torch.int64)}

    def load_optimizer_state(self, state: Dict[str, torch.Tensor]) -> None:
        self.ensure_train_state()
        for k in ("exp_avg", "exp_avg_sq"):
            t = state[k]
            if t.numel() != self.total:
                raise ValueError(f"optimizer state '{k}' has {t.numel()} elements, expected {self.total}")
            getattr(self, k).copy_(t.to(self.device, dtype=torch.float32).view(-1))
        self.opt_step = int(state["opt_step"].view(-1)[0])
        if self._step_dev is not None:
            self._step_dev.fill_(self.opt_step)

    def adamw_step_sched(self, base_lr: float, warmup_steps: int, total_steps: int, betas=(0.9, 0.999),
                         eps: float = 1e-8, weight_decay: float = 0.01, max_grad_norm: float = 1.0,
                         grad_scale: float = 1.0) -> None:
        """adamw_step with the WarmupLinear schedule and the step counter on the device (no per-step host value is a
        kernel argument, so the call can sit inside a captured HIP graph). The counter starts from self.opt_step."""
        self.ensure_train_state()
        if self._step_dev is None:
            self._step_dev = torch.tensor([self.opt_step], dtype=torch.int64, device=self.device)
        self.opt_step += 1               # host mirror (bookkeeping only; the device counter is authoritative)
        _lib.check(self.lib.qst_clip_adamw_step_sched(
            self.handle, self.params.data_ptr(), self.grads.data_ptr(), self.exp_avg.data_ptr(),
            self.exp_avg_sq.data_ptr(), base_lr, betas[0], betas[1], eps, weight_decay, max_grad_norm, grad_scale,
            int(warmup_steps), int(total_steps), self._step_dev.data_ptr(), self.grad_norm.data_ptr(),
            self._scratch.data_ptr(), _lib.current_stream_ptr()), "qst_clip_adamw_step_sched")
        self.shadow_stale = True
        self.shadow_mx_stale = True
        self.shadow_f16_stale = True
        self.shadow_f16w_stale = True


    # ------------------------------------------------------------------ mixed precision (QST_PREC_F16 training)
    def ensure_amp_scaler(self, init_scale: float = 65536.0) -> torch.Tensor:
        """The loss scaler of f16 training on the device (include/qst.h qst_amp_scaler_init): fp32 [4] {scale, growth
        tracker, last step skipped, skipped steps}. torch.cuda.amp.GradScaler's default init_scale, as ST fit(use_amp=True)."""
        if self.amp_scaler is None:
            self.amp_scaler = torch.zeros(4, dtype=torch.float32, device=self.device)
            _lib.check(self.lib.qst_amp_scaler_init(self.amp_scaler.data_ptr(), float(init_scale), _lib.current_stream_ptr()),
                       "qst_amp_scaler_init")
        return self.amp_scaler

    def adamw_step_amp(self, base_lr: float, warmup_steps: int, total_steps: int, betas=(0.9, 0.999), eps: float = 1e-8,
                       weight_decay: float = 0.01, max_grad_norm: float = 1.0, grad_scale: float = 1.0,
                       growth_factor: float = 2.0, backoff_factor: float = 0.5, growth_interval: int = 2000) -> None:
        """GradScaler.unscale_ + clip_grad_norm_ + GradScaler.step(AdamW) + update + zero_grad on the device
        (qst_clip_adamw_step_amp): the gradients in the arena carry the loss scale; a step whose gradients overflowed is
        skipped and halves the scale. total_steps <= 0: constant base_lr. No host value of the step depends on the outcome,
        so the call can sit inside a captured graph; self.opt_step counts CALLS (skipped steps are in amp_scaler[3])."""
        self.ensure_train_state()
        self.ensure_amp_scaler()
        if self._step2_dev is None:
            self._step2_dev = torch.tensor([self.opt_step, self.opt_step], dtype=torch.int64, device=self.device)
        self.opt_step += 1
        _lib.check(self.lib.qst_clip_adamw_step_amp(
            self.handle, self.params.data_ptr(), self.grads.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
            base_lr, betas[0], betas[1], eps, weight_decay, max_grad_norm, grad_scale, int(warmup_steps), int(total_steps),
            self._step2_dev.data_ptr(), self.amp_scaler.data_ptr(), growth_factor, backoff_factor, int(growth_interval),
            self.grad_norm.data_ptr(), self._scratch.data_ptr(), _lib.current_stream_ptr()), "qst_clip_adamw_step_amp")
        self.shadow_stale = True
        self.shadow_mx_stale = True
        self.shadow_f16_stale = True
        self.shadow_f16w_stale = True


def quadruplet_loss_raw(xa, xp, xq, xn, gamma, m_pn, m_pq, m_qn, p, swap, reduction: int,
                        grad_out: Optional[torch.Tensor] = None, want_grads: bool = False):
    """Direct call of qst_quadruplet_loss on contiguous fp32 CUDA tensors [B, D]."""
    lib = _lib.load()
    B, D = xa.shape
    dev = xa.device
    out = torch.empty(B if reduction == 0 else 1, dtype=torch.float32, device=dev)
    scratch = torch.empty(B, dtype=torch.float32, device=dev)
    # the four gradients are the slabs of ONE [4, B, D] buffer: viewed as [4B, D] it is the encoder backward's
    # grad_emb for the fused four-column pass, with no concatenation in between
    grads = torch.empty(4, B, D, dtype=torch.float32, device=dev).unbind(0) if want_grads else [None] * 4
    _lib.check(lib.qst_quadruplet_loss(
        xa.data_ptr(), xp.data_ptr(), xq.data_ptr(), xn.data_ptr(), B, D, gamma, m_pn, m_pq, m_qn, p, int(swap),
        reduction, out.data_ptr(), _lib.ptr(grad_out), *[_lib.ptr(g) for g in grads], scratch.data_ptr(),
        _lib.current_stream_ptr()), "qst_quadruplet_loss")
    return out, list(grads)


def stacked(grads) -> torch.Tensor:
    """The [4B, D] view behind the four gradient slabs quadruplet_loss_raw returned (no copy)."""
    base = grads[0]._base
    assert base is not None and base.dim() == 3 and all(g._base is base for g in grads)
    return base.view(-1, base.shape[-1])
