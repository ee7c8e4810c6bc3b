"""Encoder configuration and the flat parameter-arena layout shared with the C-ABI.

The reference never states these dimensions itself; it picks them by model name
(`training/main.py:114,242` -> `SentenceTransformer('all-MiniLM-L6-v2')`), so the
dims below are the published architectures named in BASELINE.json `configs`
(SURVEY.md section 8 header).

Arena layout (one fp32 buffer for params, one for grads, two for Adam moments,
one bf16 shadow for MFMA operands): every segment starts on a multiple of
ARENA_ALIGN elements so that (a) 16-byte vector access is always aligned and
(b) the fused AdamW kernel can look up weight-decay per 256-element chunk.
The order below is a contract with csrc/qst_layout.h (qst_layout_build) and is
re-derived, not copied, on the C side; tests/test_layout.py checks both agree.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Dict, List, Tuple

ARENA_ALIGN = 256  # elements

ARCH_BERT = 0
ARCH_MPNET = 1


@dataclass(frozen=True)
class EncoderConfig:
    arch: int = ARCH_BERT
    vocab_size: int = 30522
    hidden_size: int = 384
    num_layers: int = 6
    num_heads: int = 12
    intermediate_size: int = 1536
    max_position: int = 512
    type_vocab_size: int = 2
    layer_norm_eps: float = 1e-12
    normalize: bool = True          # ST `Normalize` module present (MiniLM, mpnet: yes; bare bert-base: no)
    max_seq_length: int = 256       # ST `Transformer.max_seq_length`
    rel_buckets: int = 32           # MPNet only
    rel_max_distance: int = 128     # MPNet only
    pad_token_id: int = 0           # BERT 0, MPNet 1

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    def to_dict(self) -> dict:
        return asdict(self)


PRESETS: Dict[str, EncoderConfig] = {
    # MiniLM-L6 H=384, 6 layers, 12 heads, I=1536, V=30522, P=512, T=2, eps 1e-12
    "all-MiniLM-L6-v2": EncoderConfig(),
    # mpnet-base: H=768, 12 layers, 12 heads, I=3072, V=30527, P=514, eps 1e-5, +32x12 rel-bias
    "all-mpnet-base-v2": EncoderConfig(
        arch=ARCH_MPNET, vocab_size=30527, hidden_size=768, num_layers=12, num_heads=12,
        intermediate_size=3072, max_position=514, type_vocab_size=0, layer_norm_eps=1e-5,
        normalize=True, max_seq_length=384, pad_token_id=1),
    # bare bert-base-uncased: ST auto-wraps Transformer + mean Pooling, no Normalize
    "bert-base-uncased": EncoderConfig(
        hidden_size=768, num_layers=12, num_heads=12, intermediate_size=3072,
        normalize=False, max_seq_length=512),
    # tiny configs used by the parity fixtures (tests/golden/tiny_*.npz)
    "tiny-bert": EncoderConfig(
        vocab_size=128, hidden_size=64, num_layers=2, num_heads=2, intermediate_size=256,
        max_position=64, max_seq_length=64),
    "tiny-mpnet": EncoderConfig(
        arch=ARCH_MPNET, vocab_size=128, hidden_size=128, num_layers=2, num_heads=2,
        intermediate_size=256, max_position=66, type_vocab_size=0, layer_norm_eps=1e-5,
        max_seq_length=64, pad_token_id=1),
    # MiniLM layer dimensions on two layers and a 4,096-word vocabulary: the fixture that reaches the flagship kernels
    # (GEMM + LayerNorm fused, 8-range grouped wgrad, single-workgroup attention backward: M >= 16384 token rows) while
    # staying cheap for the HF reference on the CPU (tests/golden/encoder_golden.npz: minilm2l_fused)
    "minilm-2l": EncoderConfig(num_layers=2, vocab_size=4096),
}


def _align(n: int) -> int:
    return (n + ARENA_ALIGN - 1) // ARENA_ALIGN * ARENA_ALIGN


@dataclass(frozen=True)
class Segment:
    name: str            # build-internal name, e.g. "layer.3.w_qkv"
    offset: int          # element offset into the arena
    shape: Tuple[int, ...]
    decay: bool          # ST fit(): weight decay on everything but bias / LayerNorm.*
    gemm: bool           # has a bf16 shadow used as an MFMA operand

    @property
    def numel(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n


def build_layout(cfg: EncoderConfig) -> Tuple[List[Segment], int]:
    """Return (segments, total_elements). Mirrors qst_layout_build() in csrc/qst_layout.h."""
    H, I, N = cfg.hidden_size, cfg.intermediate_size, cfg.num_layers
    segs: List[Segment] = []
    off = 0

    def add(name, shape, decay, gemm=False):
        nonlocal off
        s = Segment(name, off, tuple(shape), decay, gemm)
        segs.append(s)
        off = _align(off + s.numel)

    add("word_emb", (cfg.vocab_size, H), True)
    add("pos_emb", (cfg.max_position, H), True)
    if cfg.type_vocab_size > 0:
        add("type_emb", (cfg.type_vocab_size, H), True)
    add("emb_ln_g", (H,), False)
    add("emb_ln_b", (H,), False)
    if cfg.arch == ARCH_MPNET:
        add("rel_bias", (cfg.rel_buckets, cfg.num_heads), True)
    for l in range(N):
        p = f"layer.{l}."
        add(p + "w_qkv", (3 * H, H), True, True)
        add(p + "b_qkv", (3 * H,), False)
        add(p + "w_o", (H, H), True, True)
        add(p + "b_o", (H,), False)
        add(p + "ln1_g", (H,), False)
        add(p + "ln1_b", (H,), False)
        add(p + "w_1", (I, H), True, True)
        add(p + "b_1", (I,), False)
        add(p + "w_2", (H, I), True, True)
        add(p + "b_2", (H,), False)
        add(p + "ln2_g", (H,), False)
        add(p + "ln2_b", (H,), False)
    return segs, off


def hf_param_views(cfg: EncoderConfig) -> List[Tuple[str, str, int, Tuple[int, ...]]]:
    """Map HF/ST parameter names onto arena slices.

    Returns [(hf_name, segment_name, element_offset_within_segment, shape)].
    Names are what `SentenceTransformer.named_parameters()` yields below the
    `0.auto_model.` prefix for BertModel / MPNetModel, so
    ST's `fit()` weight-decay name filter ('bias', 'LayerNorm.bias',
    'LayerNorm.weight') behaves identically.
    """
    H, I = cfg.hidden_size, cfg.intermediate_size
    out = [("embeddings.word_embeddings.weight", "word_emb", 0, (cfg.vocab_size, H)),
           ("embeddings.position_embeddings.weight", "pos_emb", 0, (cfg.max_position, H))]
    if cfg.type_vocab_size > 0:
        out.append(("embeddings.token_type_embeddings.weight", "type_emb", 0, (cfg.type_vocab_size, H)))
    out += [("embeddings.LayerNorm.weight", "emb_ln_g", 0, (H,)),
            ("embeddings.LayerNorm.bias", "emb_ln_b", 0, (H,))]
    for l in range(cfg.num_layers):
        s = f"layer.{l}."
        if cfg.arch == ARCH_BERT:
            p = f"encoder.layer.{l}."
            qkv = [p + "attention.self.query", p + "attention.self.key", p + "attention.self.value"]
            o, ln1 = p + "attention.output.dense", p + "attention.output.LayerNorm"
        else:
            p = f"encoder.layer.{l}."
            qkv = [p + "attention.attn.q", p + "attention.attn.k", p + "attention.attn.v"]
            o, ln1 = p + "attention.attn.o", p + "attention.LayerNorm"
        for j, nm in enumerate(qkv):
            out.append((nm + ".weight", s + "w_qkv", j * H * H, (H, H)))
            out.append((nm + ".bias", s + "b_qkv", j * H, (H,)))
        out += [(o + ".weight", s + "w_o", 0, (H, H)), (o + ".bias", s + "b_o", 0, (H,)),
                (ln1 + ".weight", s + "ln1_g", 0, (H,)), (ln1 + ".bias", s + "ln1_b", 0, (H,)),
                (p + "intermediate.dense.weight", s + "w_1", 0, (I, H)),
                (p + "intermediate.dense.bias", s + "b_1", 0, (I,)),
                (p + "output.dense.weight", s + "w_2", 0, (H, I)),
                (p + "output.dense.bias", s + "b_2", 0, (H,)),
                (p + "output.LayerNorm.weight", s + "ln2_g", 0, (H,)),
                (p + "output.LayerNorm.bias", s + "ln2_b", 0, (H,))]
    if cfg.arch == ARCH_MPNET:
        out.append(("encoder.relative_attention_bias.weight", "rel_bias", 0, (cfg.rel_buckets, cfg.num_heads)))
    return out


def forward_flops_per_sequence(cfg: EncoderConfig, L: int) -> float:
    """N_layers * L * (8H^2 + 4HI + 4LH)."""
    H, I = cfg.hidden_size, cfg.intermediate_size
    return float(cfg.num_layers) * L * (8.0 * H * H + 4.0 * H * I + 4.0 * L * H)
