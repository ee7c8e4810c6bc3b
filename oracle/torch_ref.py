"""TEST INFRASTRUCTURE ONLY -- fp32 torch restatement of the hot path (CPU oracle).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file. The product path (quadruplet-sentence-transformer_amd/) never does.

What it restates, op for op, with plain torch primitives (no nn.Module, no
transformers import), so autograd gives the backward oracle as well:

  * BertModel.forward  (transformers 4.30.2 per requirements.txt:9; local copy
    transformers 5.15.0 modeling_bert.py:53-108 embeddings, :111-136 eager
    attention, :282-293 / :325-351 output blocks, :354-416 layer)
  * MPNetModel.forward (modeling_mpnet.py:58-95 embeddings, :115-174 attention,
    :312-348 relative bias, :873-881 position ids)
  * sentence-transformers 2.2.2 Pooling(mean) + Normalize head (third-party, not
    under /root/reference; call site models/quadruplet_sentence_transformer.py:42-60;
    arithmetic in SURVEY.md 8a row a4)
  * gamma_quadruplet_loss (/root/reference/models/losses/losses.py:9-69) on top of
    torch's triplet_margin_loss / pairwise_distance semantics (eps added to every
    component of the difference)

Pinned by tests/golden/*.npz, which oracle/make_golden.py generates from the real
reference loss module and HF BertModel/MPNetModel (tests/test_oracle_golden.py).

`bf16_operands=True` (or "bf16"; "f16" = IEEE half, the operand type of QST_PREC_F16 -- operand_dtype below) rounds every
GEMM operand (activations, weights, Q/K/V, P) to that 16-bit type and keeps fp32 results -- the arithmetic the HIP kernels do -- so
kernel bugs can be told apart from bf16 rounding. In that mode the contractions are
accumulated in fp64 and rounded to fp32 once: products of bf16 values are exact in
fp64 and the sum no longer depends on the BLAS library's reduction order or thread
count, so the oracle gives the same numbers on every box (a value that sits on a
bf16 rounding boundary used to flip with the host's BLAS, and the test bounds chased it).
The BACKWARD of that mode rounds what the kernels round: the gradient entering every product (dY of a Linear, dO / dS of the
attention products) is a bf16 operand there too, and bias gradients are column sums of the rounded dY (_MmBf16, _LinearBf16).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F


def _r(x: torch.Tensor, on: bool) -> torch.Tensor:
    """Round-trip through bf16 with a straight-through gradient."""
    if not on:
        return x
    return x + (x.detach().to(torch.bfloat16).to(torch.float32) - x.detach())


_OPERAND_DTYPES = {True: torch.bfloat16, "bf16": torch.bfloat16, "f16": torch.float16, "fp16": torch.float16}


def operand_dtype(bf16_operands):
    """The 16-bit matrix-core operand type a `bf16_operands` argument names: False / None = none (plain fp32), True / "bf16" =
    bfloat16 (QST_PREC_BF16), "f16" = IEEE half (QST_PREC_F16: 11 significand bits, the reference's `use_amp` precision class,
    /root/reference/training/main.py:142). torch's float32 -> float16 cast rounds to nearest even and overflows to inf; the
    kernels' forward epilogues saturate at 65,504 instead -- no golden case comes near either."""
    if bf16_operands is None or bf16_operands is False:
        return None
    if isinstance(bf16_operands, torch.dtype):
        return bf16_operands
    return _OPERAND_DTYPES[bf16_operands]


def _b16(x: torch.Tensor, dt: torch.dtype = torch.bfloat16) -> torch.Tensor:
    return x.detach().to(dt).to(torch.float64)


class _MmBf16(torch.autograd.Function):
    """a @ b as the HIP path multiplies: both operands rounded to bf16, exact products, fp64 accumulation, one rounding to
    fp32 (independent of the host's BLAS) -- and the SAME in the backward, where the incoming gradient is a bf16 matrix-core
    operand too: dA = bf16(g) @ bf16(b)^T, dB = bf16(a)^T @ bf16(g) (the attention backward's dP, dV, dQ, dK products;
    csrc/attention.hip rounds dO, P and dS once each)."""

    @staticmethod
    def forward(ctx, a, b, dt=torch.bfloat16):
        a16, b16 = _b16(a, dt), _b16(b, dt)
        ctx.save_for_backward(a16, b16)
        ctx.shapes = (a.shape, b.shape)
        ctx.dt = dt
        return torch.matmul(a16, b16).float()

    @staticmethod
    def backward(ctx, g):
        a16, b16 = ctx.saved_tensors
        g16 = _b16(g, ctx.dt)
        ga = torch.matmul(g16, b16.transpose(-1, -2)).sum_to_size(ctx.shapes[0]).float()
        gb = torch.matmul(a16.transpose(-1, -2), g16).sum_to_size(ctx.shapes[1]).float()
        return ga, gb, None


class _LinearBf16(torch.autograd.Function):
    """x @ w^T + b on bf16-rounded operands (fp64 accumulation); backward as the dgrad / wgrad kernels take it: dY rounded
    to bf16 once, dX = bf16(dY) @ bf16(W), dW = bf16(dY)^T @ bf16(X), db = the column sums of the bf16-ROUNDED dY (the wgrad
    kernel sums the fragments it already holds, csrc/gemm.hip gemm_tn_group_kernel)."""

    @staticmethod
    def forward(ctx, x, w, b, dt=torch.bfloat16):
        x16, w16 = _b16(x, dt), _b16(w, dt)
        ctx.save_for_backward(x16, w16)
        ctx.has_bias = b is not None
        ctx.dt = dt
        y = torch.matmul(x16, w16.t()).float()
        return y if b is None else y + b

    @staticmethod
    def backward(ctx, g):
        x16, w16 = ctx.saved_tensors
        g16 = _b16(g, ctx.dt)
        g2, x2 = g16.reshape(-1, g16.shape[-1]), x16.reshape(-1, x16.shape[-1])
        return torch.matmul(g16, w16).float(), torch.matmul(g2.t(), x2).float(), (g2.sum(0).float() if ctx.has_bias else None), None


def _mm(a, b, bf16):
    """a @ b; with bf16 operands: _MmBf16 (forward and backward on bf16-rounded operands)."""
    dt = operand_dtype(bf16)
    if dt is None:
        return torch.matmul(a, b)
    return _MmBf16.apply(a, b, dt)


def _linear(x, w, b, bf16):
    dt = operand_dtype(bf16)
    if dt is None:
        return F.linear(x, w, b)
    return _LinearBf16.apply(x, w, b, dt)


def mpnet_position_ids(ids: torch.Tensor, pad: int = 1) -> torch.Tensor:
    m = ids.ne(pad).int()
    return (torch.cumsum(m, dim=1).type_as(m) * m).long() + pad


def mpnet_bucket_table(L: int, num_buckets: int = 32, max_distance: int = 128) -> torch.Tensor:
    """[L, L] int64 bucket of (j - i); modeling_mpnet.py:329-348."""
    ctx = torch.arange(L)[:, None]
    mem = torch.arange(L)[None, :]
    n = -(mem - ctx)
    nb = num_buckets // 2
    ret = (n < 0).long() * nb
    n = n.abs()
    max_exact = nb // 2
    is_small = n < max_exact
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact)
                         * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return ret + torch.where(is_small, n, large)


def encoder_forward(P: Dict[str, torch.Tensor], cfg, ids: torch.Tensor, mask: torch.Tensor,
                    type_ids: Optional[torch.Tensor] = None, bf16_operands: bool = False,
                    collect: Optional[list] = None, dropout=None) -> torch.Tensor:
    """Token embeddings [n, L, H]. P is keyed by the build's segment names (config.build_layout).
    dropout: None (eval mode) or an oracle/dropout_ref.Masks -- train() mode with GIVEN masks at HF's four places:
    BertEmbeddings (after its LayerNorm), BertSelfAttention (on the probabilities), BertSelfOutput and BertOutput (on the
    dense output, before the residual add); MPNet has the same four (modeling_mpnet.py)."""
    n, L = ids.shape
    H, A = cfg.hidden_size, cfg.num_heads
    d = H // A
    bf = bf16_operands
    if cfg.arch == 0:
        x = P["word_emb"][ids]
        if cfg.type_vocab_size > 0:
            tt = type_ids if type_ids is not None else torch.zeros_like(ids)
            x = x + P["type_emb"][tt]                      # HF order: (word + type) + position
        x = x + P["pos_emb"][torch.arange(L)][None]
        rel = None
    else:
        x = P["word_emb"][ids] + P["pos_emb"][mpnet_position_ids(ids, cfg.pad_token_id)]
        bucket = mpnet_bucket_table(L, cfg.rel_buckets, cfg.rel_max_distance)
        rel = P["rel_bias"][bucket].permute(2, 0, 1)[None]          # [1, A, L, L]
    x = F.layer_norm(x, (H,), P["emb_ln_g"], P["emb_ln_b"], cfg.layer_norm_eps)
    if dropout is not None:
        x = x * dropout.embed(n, L, H)
    if collect is not None:
        collect.append(x)
    neg = torch.finfo(torch.float32).min
    add_mask = (1.0 - mask[:, None, None, :].to(torch.float32)) * neg   # modeling_bert.py:688-708
    for l in range(cfg.num_layers):
        p = f"layer.{l}."
        qkv = _linear(x, P[p + "w_qkv"], P[p + "b_qkv"], bf)
        q, k, v = [t.view(n, L, A, d).transpose(1, 2) for t in qkv.split(H, dim=-1)]
        s = _mm(q, k.transpose(-1, -2), bf) / math.sqrt(d)
        if rel is not None:
            s = s + rel
        s = s + add_mask
        pr = torch.softmax(s, dim=-1)
        if dropout is not None:
            pr = pr * dropout.probs(l, n, A, L)
        ctx = _mm(pr, v, bf).transpose(1, 2).reshape(n, L, H)
        a = _linear(ctx, P[p + "w_o"], P[p + "b_o"], bf)
        if dropout is not None:
            a = a * dropout.attn_out(l, n, L, H)
        x = F.layer_norm(a + x, (H,), P[p + "ln1_g"], P[p + "ln1_b"], cfg.layer_norm_eps)
        h = F.gelu(_linear(x, P[p + "w_1"], P[p + "b_1"], bf))          # erf GELU
        o = _linear(h, P[p + "w_2"], P[p + "b_2"], bf)
        if dropout is not None:
            o = o * dropout.ffn_out(l, n, L, H)
        x = F.layer_norm(o + x, (H,), P[p + "ln2_g"], P[p + "ln2_b"], cfg.layer_norm_eps)
        if collect is not None:
            collect.append(x)
    return x


def st_head(tok: torch.Tensor, mask: torch.Tensor, normalize: bool) -> torch.Tensor:
    """ST Pooling(mean) [+ Normalize]: sum(tok*mask)/clamp(sum mask, 1e-9); F.normalize eps 1e-12."""
    m = mask[:, :, None].to(tok.dtype)
    e = (tok * m).sum(1) / m.sum(1).clamp(min=1e-9)
    if normalize:
        e = e / e.norm(p=2, dim=1, keepdim=True).clamp_min(1e-12)
    return e


def sentence_embeddings(P, cfg, ids, mask, type_ids=None, bf16_operands=False, dropout=None):
    return st_head(encoder_forward(P, cfg, ids, mask, type_ids, bf16_operands, dropout=dropout), mask, cfg.normalize)


def _pdist(x1, x2, p, eps=1e-6):
    # torch.nn.functional.pairwise_distance: ||x1 - x2 + eps||_p over the last dim
    return torch.linalg.vector_norm(x1 - x2 + eps, ord=p, dim=-1)


def _tml(a, pos, neg, margin, p, swap):
    dp = _pdist(a, pos, p)
    dn = _pdist(a, neg, p)
    if swap:
        dn = torch.minimum(dn, _pdist(pos, neg, p))
    return torch.clamp_min(margin + dp - dn, 0.0)


def gamma_quadruplet_loss_ref(xa, xp, xq, xn, gamma=0.6, margin_pos_neg=1.0, margin_pos_part=0.5,
                              margin_part_neg=0.5, p=2.0, swap=False, reduction="mean"):
    """Restatement of /root/reference/models/losses/losses.py:35-69 (same term order)."""
    a = _tml(xa, xp, xn, margin_pos_neg, p, swap)       # :35-43
    b = _tml(xa, xq, xn, margin_part_neg, p, swap)      # :44-52
    c = _tml(xa, xp, xq, margin_pos_part, p, swap)      # :53-61
    if reduction == "none":
        return a + gamma * b + (1 - gamma) * c
    if reduction == "sum":
        return a.sum() + (gamma * b).sum() + ((1 - gamma) * c).sum()
    return a.mean() + (gamma * b).mean() + ((1 - gamma) * c).mean()


def arena_to_dict(arena, cfg, requires_grad: bool = False) -> Dict[str, torch.Tensor]:
    """Split a flat fp32 arena (numpy or torch) into named leaf tensors."""
    import importlib
    qst = importlib.import_module("quadruplet_sentence_transformer_amd.config")
    segs, _ = qst.build_layout(cfg)
    t = torch.as_tensor(arena, dtype=torch.float32)
    out = {}
    for s in segs:
        v = t[s.offset:s.offset + s.numel].clone().view(*s.shape)
        v.requires_grad_(requires_grad)
        out[s.name] = v
    return out


def quadruplet_step(P, cfg, ids4, mask4, types4=None, loss_kw=None, bf16_operands=False, dropout=None):
    """Forward of one quadruplet batch: ids4 [4,B,L] -> (loss, emb [4,B,H]).

    Column order = reference/positive/part_positive/negative
    (/root/reference/models/quadruplet_sentence_transformer.py:24-60).
    """
    loss_kw = loss_kw or {}
    four, B, L = ids4.shape
    ids = ids4.reshape(4 * B, L)
    mask = mask4.reshape(4 * B, L)
    tt = types4.reshape(4 * B, L) if types4 is not None else None
    emb = sentence_embeddings(P, cfg, ids, mask, tt, bf16_operands, dropout).view(4, B, -1)
    loss = gamma_quadruplet_loss_ref(emb[0], emb[1], emb[2], emb[3], **loss_kw)
    return loss, emb


def mx_quant(x: torch.Tensor):
    """MXFP8 (OCP e4m3 elements, one E8M0 power-of-two scale per 32 consecutive elements of the last dimension) as
    libqst quantises it (csrc/gemm.hip mx_exponent / quant_mx_kernel): per block, e = the smallest exponent with
    amax * 2^-e <= 448 -- from amax's own exponent E and mantissa: e = E - 8, +1 when the mantissa exceeds 1.75 --
    clamped to [-127, 126], -127 for an all-zero block; elements = round-to-nearest-even(x * 2^-e) to e4m3.
    Returns (q uint8 [..., K], scale uint8 [..., K/32] = e + 127, dequantised fp32 [..., K])."""
    K = x.shape[-1]
    assert K % 32 == 0
    xb = x.detach().to(torch.float32).reshape(-1, K // 32, 32)
    amax = xb.abs().amax(-1)
    bits = amax.contiguous().view(torch.int32)
    E = ((bits >> 23) & 0xFF) - 127
    e = E - 8 + ((bits & 0x7FFFFF) > 0x600000).to(torch.int32)
    e = torch.where(((bits >> 23) & 0xFF) == 0, torch.full_like(e, -127), e.clamp(-127, 126))
    scaled = torch.ldexp(xb, (-e)[..., None])
    q8 = scaled.to(torch.float8_e4m3fn)
    deq = torch.ldexp(q8.to(torch.float32), e[..., None]).reshape(x.shape)
    return q8.view(torch.uint8).reshape(x.shape), (e + 127).to(torch.uint8).reshape(*x.shape[:-1], K // 32), deq


# Accumulation of the MXFP8 oracle's products: False = fp32 through the host BLAS (what rounds 2-4 ran), True = fp64 (order-free).
# The two differ by ~1e-7 relative before the NEXT quantisation -- enough to flip an e4m3 rounding (one step = 2^-3 relative) of a
# value that sits on a boundary; tools/fuzz_shapes.py ... fp8train prints the HIP path's distance to either (DESIGN.md finding 37).
MX_ACC64 = False


def _mx_linear(a: torch.Tensor, w: torch.Tensor, b):
    if not MX_ACC64:
        return F.linear(a, w, b)
    y = torch.matmul(a.double(), w.double().t()).float()
    return y if b is None else y + b


def _mx(x: torch.Tensor, via_bf16: bool) -> torch.Tensor:
    """Operand of an MXFP8 GEMM: optionally rounded to bf16 first (activations that reach the quantiser as bf16 tensors)."""
    if via_bf16:
        x = x.to(torch.bfloat16).to(torch.float32)
    return mx_quant(x)[2]


class _LinearMxFwdBf16Bwd(torch.autograd.Function):
    """One Linear of the fp8 TRAINING path (csrc/qst_api.hip forward_mx_train + the bf16 backward): the forward product is
    taken on MXFP8 operands, the backward is the bf16 path's -- dX = bf16(dY) . bf16(W), dW = bf16(dY)^T . bf16(X), db = column
    sums of the bf16-rounded dY -- on the UNquantised operands (fp32 master weights; the bf16 copy of the activation)."""

    @staticmethod
    def forward(ctx, a, w, b, a_via_bf16):
        ctx.save_for_backward(a.to(torch.bfloat16).to(torch.float32), w.to(torch.bfloat16).to(torch.float32))
        return _mx_linear(_mx(a, a_via_bf16), _mx(w, False), b)

    @staticmethod
    def backward(ctx, g):
        a16, w16 = ctx.saved_tensors
        g16 = g.to(torch.bfloat16).to(torch.float32)
        g2, a2 = g16.reshape(-1, g16.shape[-1]), a16.reshape(-1, a16.shape[-1])
        return (g16.double() @ w16.double()).float(), (g2.double().t() @ a2.double()).float(), g2.sum(0), None


def encoder_forward_mx(P: Dict[str, torch.Tensor], cfg, ids: torch.Tensor, mask: torch.Tensor,
                       type_ids: Optional[torch.Tensor] = None, train: bool = False, dropout=None) -> torch.Tensor:
    """QST_PREC_FP8 oracle (inference): encoder_forward with every Linear computed on MXFP8 operands -- weights
    quantised from fp32; the layer input, the attention output and the LayerNorm-1 output quantised from their bf16
    copies; gelu(u) quantised from fp32 (it never exists in another format) -- and attention on bf16 operands, as the
    HIP pipeline does (csrc/qst_api.hip forward_mx). train=True: the fp8 TRAINING forward (forward_mx_train) with the bf16
    path's backward attached to every Linear (_LinearMxFwdBf16Bwd) -- autograd through the result is the oracle of
    "fp8 forward GEMMs, bf16 dgrad / wgrad". dropout (training only): an oracle/dropout_ref.Masks, applied at the four
    places encoder_forward applies it."""
    n, L = ids.shape
    H, A = cfg.hidden_size, cfg.num_heads
    d = H // A
    if cfg.arch == 0:
        x = P["word_emb"][ids]
        if cfg.type_vocab_size > 0:
            tt = type_ids if type_ids is not None else torch.zeros_like(ids)
            x = x + P["type_emb"][tt]
        x = x + P["pos_emb"][torch.arange(L)][None]
        rel = None
    else:
        x = P["word_emb"][ids] + P["pos_emb"][mpnet_position_ids(ids, cfg.pad_token_id)]
        bucket = mpnet_bucket_table(L, cfg.rel_buckets, cfg.rel_max_distance)
        rel = P["rel_bias"][bucket].permute(2, 0, 1)[None]
    x = F.layer_norm(x, (H,), P["emb_ln_g"], P["emb_ln_b"], cfg.layer_norm_eps)
    if dropout is not None:
        x = x * dropout.embed(n, L, H)
    neg = torch.finfo(torch.float32).min
    add_mask = (1.0 - mask[:, None, None, :].to(torch.float32)) * neg

    def lin(a, w, b, via_bf16):
        if train:
            return _LinearMxFwdBf16Bwd.apply(a, w, b, via_bf16)
        return _mx_linear(_mx(a, via_bf16), _mx(w, False), b)
    for l in range(cfg.num_layers):
        p = f"layer.{l}."
        qkv = _r(lin(x, P[p + "w_qkv"], P[p + "b_qkv"], True), True)            # the QKV GEMM writes bf16
        q, k, v = [t.view(n, L, A, d).transpose(1, 2) for t in qkv.split(H, dim=-1)]
        s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(d)
        if rel is not None:
            s = s + rel
        s = s + add_mask
        pr = torch.softmax(s, dim=-1)
        if dropout is not None:
            pr = pr * dropout.probs(l, n, A, L)
        ctx = torch.matmul(_r(pr, True), v).transpose(1, 2).reshape(n, L, H)
        a = lin(ctx, P[p + "w_o"], P[p + "b_o"], True)
        if dropout is not None:
            a = a * dropout.attn_out(l, n, L, H)
        x = F.layer_norm(a + x, (H,), P[p + "ln1_g"], P[p + "ln1_b"], cfg.layer_norm_eps)
        h = F.gelu(lin(x, P[p + "w_1"], P[p + "b_1"], True))
        o = lin(h, P[p + "w_2"], P[p + "b_2"], train)        # (training keeps gelu(u) as bf16 and quantises that copy)
        if dropout is not None:
            o = o * dropout.ffn_out(l, n, L, H)
        x = F.layer_norm(o + x, (H,), P[p + "ln2_g"], P[p + "ln2_b"], cfg.layer_norm_eps)
    return x
