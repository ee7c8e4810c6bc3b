"""Counter-based synthetic weights and quadruplet batches.

There are no pretrained checkpoints or vocab files offline, so benchmark and
parity inputs are synthetic. Everything here is a pure function of
(seed, tensor index, element index) built from integer hashing and exact
dyadic arithmetic: no libm call, so the numbers are bit-identical in this
container and on the GPU box regardless of numpy's SIMD dispatch.

Weights follow HF's init shape (`BertPreTrainedModel._init_weights`):
Linear/Embedding ~ zero-mean, std 0.02 (Irwin-Hall n=4 instead of a true
normal), LayerNorm gamma=1 beta=0, biases 0. `scale`/`bias_scale` let parity
tests use "trained-like" magnitudes where every term of the network matters.
"""
from __future__ import annotations

import numpy as np

from .config import EncoderConfig, build_layout

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + _G).astype(np.uint64)
        x = (x ^ (x >> np.uint64(30))) * _M1
        x = (x ^ (x >> np.uint64(27))) * _M2
        return x ^ (x >> np.uint64(31))


def hash_u64(seed: int, stream: int, n: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream)], dtype=np.uint64))[0]
        return _splitmix64(np.arange(n, dtype=np.uint64) + base)


def approx_normal(seed: int, stream: int, n: int, std: float) -> np.ndarray:
    """Irwin-Hall(4) from the four 16-bit fields of one 64-bit hash: mean 0, variance 1, exact in fp64."""
    h = hash_u64(seed, stream, n)
    s = np.zeros(n, dtype=np.float64)
    for k in range(4):
        s += ((h >> np.uint64(16 * k)) & np.uint64(0xFFFF)).astype(np.float64)
    # each field is U{0..65535}: mean 32767.5, var (65536^2-1)/12
    s = (s - 4 * 32767.5) / np.sqrt(4 * (65536.0 ** 2 - 1) / 12.0)
    return (s * std).astype(np.float32)


def uniform_int(seed: int, stream: int, n: int, lo: int, hi: int) -> np.ndarray:
    """Integers in [lo, hi) (modulo bias is irrelevant here)."""
    h = hash_u64(seed, stream, n)
    return (lo + (h % np.uint64(hi - lo)).astype(np.int64)).astype(np.int64)


def synthetic_params(cfg: EncoderConfig, seed: int = 14, std: float = 0.02,
                     bias_std: float = 0.0, ln_jitter: float = 0.0) -> np.ndarray:
    """Flat fp32 parameter arena (layout: config.build_layout)."""
    segs, total = build_layout(cfg)
    arena = np.zeros(total, dtype=np.float32)
    for idx, s in enumerate(segs):
        leaf = s.name.split(".")[-1]
        if leaf in ("emb_ln_g", "ln1_g", "ln2_g"):
            v = np.ones(s.numel, dtype=np.float32)
            if ln_jitter:
                v += approx_normal(seed, 1000 + idx, s.numel, ln_jitter)
        elif leaf in ("emb_ln_b", "ln1_b", "ln2_b"):
            v = approx_normal(seed, 1000 + idx, s.numel, ln_jitter) if ln_jitter else np.zeros(s.numel, np.float32)
        elif leaf.startswith("b_"):
            v = approx_normal(seed, 1000 + idx, s.numel, bias_std) if bias_std else np.zeros(s.numel, np.float32)
        else:
            v = approx_normal(seed, 1000 + idx, s.numel, std)
        arena[s.offset:s.offset + s.numel] = v
    return arena


def synthetic_quadruplets(cfg: EncoderConfig, batch: int, seq_len: int, seed: int = 14,
                          ragged: bool = False, step: int = 0, rank: int = 0):
    """ids/mask/type_ids as int64 [4, batch, seq_len] (SURVEY.md 8d synthetic inputs).

    ids ~ U{lo..V-1} with lo = min(1000, V/4); full-length masks for throughput
    runs, per-sequence lengths ~ U{L/8..L} (zero-padded) when ragged.
    """
    n = 4 * batch * seq_len
    stream = 7 + 1000003 * step + 7919 * rank
    lo = min(1000, cfg.vocab_size // 4)
    ids = uniform_int(seed, stream, n, lo, cfg.vocab_size).reshape(4, batch, seq_len)
    mask = np.ones((4, batch, seq_len), dtype=np.int64)
    if ragged:
        lens = uniform_int(seed, stream + 1, 4 * batch, max(1, seq_len // 8), seq_len + 1).reshape(4, batch)
        pos = np.arange(seq_len)[None, None, :]
        mask = (pos < lens[:, :, None]).astype(np.int64)
        ids = np.where(mask == 1, ids, cfg.pad_token_id)
    types = np.zeros_like(ids)
    return ids, mask, types


def mask_edge_cases(ids: np.ndarray, mask: np.ndarray, pad_id: int):
    """Attention-mask shapes a tokenizer never produces but callers can: given a [4, B>=3, L>=64] batch, make
    sequence (0, 1) all padding (HF: every key carries finfo.min -> uniform attention; ST: mask sum clamped at 1e-9 ->
    zero embedding), (1, 0) LEFT-padded by 40 positions (whole leading 32-key tile masked) and (2, 2) left-padded by
    exactly one 32-key tile. Returns modified copies."""
    ids, mask = ids.copy(), mask.copy()
    L = mask.shape[2]
    mask[0, 1, :] = 0
    mask[1, 0, :40] = 0
    mask[1, 0, 40:] = 1
    mask[2, 2, :32] = 0
    mask[2, 2, 32:L - 3] = 1
    ids = np.where(mask == 1, np.where(ids == pad_id, pad_id + 7, ids), pad_id)
    return ids, mask
