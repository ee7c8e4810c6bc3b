"""TEST INFRASTRUCTURE ONLY -- numpy restatement of libqst's dropout masks (include/qst_kernels.h: QstDrop).

The reference trains with torch's stateful Philox dropout inside HF BertModel / MPNetModel (train() mode, HF defaults
hidden_dropout_prob = attention_probs_dropout_prob = 0.1; /root/reference/training/main.py:128 via SentenceTransformer.fit);
its random stream is not part of any contract, only the distribution is: every element dropped independently with
probability p, kept elements scaled by 1 / (1 - p). libqst's masks are counter-based -- a pure function of (seed, step,
site, element index), so backward recomputes them instead of storing them -- and this module regenerates them bit for bit,
so that the torch oracle can run WITH THE SAME MASKS and the usual forward / gradient tolerances apply. The distribution
itself is checked in tests/test_dropout_host.py. Only tests/ may import this module.

    hash32(x):  x ^= x >> 16; x *= 0x7feb352d; x ^= x >> 15; x *= 0x846ca68b; x ^= x >> 16          (uint32)
    key       = hash32(seed_lo ^ hash32(step * 0x9E3779B9 + site) ^ rotl16(seed_hi))
    element i: 16 bits = the low (i even) / high (i odd) half of hash32((i >> 1) ^ key); dropped iff bits < thr16,
               thr16 = round(p * 65536); multiplier of a kept element = 65536 / (65536 - thr16)
    attention probabilities (the largest masked tensor, rebuilt in three kernels) use the cheaper 8-bit form: byte i & 3 of
               hash32((i >> 2) ^ key), dropped iff byte < thr8 = min(255, (thr16 + 128) >> 8), multiplier 256 / (256 - thr8):
               p = 0.1 becomes 26/256 = 0.1016 with the scale that matches it.
"""
import numpy as np

SITE_EMBED = 0xE0


def site_attn_out(layer):
    return 4 * layer + 0


def site_ffn_out(layer):
    return 4 * layer + 1


def site_probs(layer):
    return 4 * layer + 2


def hash32(x):
    x = np.asarray(x, dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
    return x


def thr16_of(p):
    return min(65535, int(np.floor(np.float32(p) * np.float32(65536.0) + np.float32(0.5))))


def mask_key(seed, step, site):
    lo, hi = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        inner = hash32(np.uint32(step) * np.uint32(0x9E3779B9) + np.uint32(site))
    rot = np.uint32(((int(hi) << 16) | (int(hi) >> 16)) & 0xFFFFFFFF)
    return hash32(lo ^ inner ^ rot)


def multipliers(seed, step, site, n, p):
    """float32 [n]: 0 for dropped elements, 65536 / (65536 - thr16) for kept ones."""
    thr = thr16_of(p)
    if thr == 0:
        return np.ones(n, dtype=np.float32)
    i = np.arange(n, dtype=np.uint64)
    h = hash32((i >> np.uint64(1)).astype(np.uint32) ^ mask_key(seed, step, site))
    bits = np.where((i & np.uint64(1)) == 0, h & np.uint32(0xFFFF), h >> np.uint32(16))
    scale = np.float32(65536.0) / np.float32(65536 - thr)
    return np.where(bits >= thr, scale, np.float32(0.0)).astype(np.float32)


def thr8_of(p):
    return min(255, (thr16_of(p) + 128) >> 8)


def multipliers8(seed, step, site, n, p):
    """The 8-bit / four-elements-per-word form used for attention probabilities."""
    thr = thr8_of(p)
    if thr16_of(p) == 0 or thr == 0:
        return np.ones(n, dtype=np.float32)
    i = np.arange(n, dtype=np.uint64)
    h = hash32((i >> np.uint64(2)).astype(np.uint32) ^ mask_key(seed, step, site))
    byte = (h >> (np.uint32(8) * (i & np.uint64(3)).astype(np.uint32))) & np.uint32(0xFF)
    scale = np.float32(256.0) / np.float32(256 - thr)
    return np.where(byte >= thr, scale, np.float32(0.0)).astype(np.float32)


class Masks:
    """The masks of one training step, in the shapes oracle/torch_ref.encoder_forward multiplies by.
    step = the number of training forwards the HIP handle has run, this one included (the counter advances first)."""

    def __init__(self, seed, step, p_hidden, p_attn):
        self.seed, self.step, self.p_hidden, self.p_attn = seed, step, p_hidden, p_attn

    def _t(self, site, shape, p):
        import torch
        return torch.from_numpy(multipliers(self.seed, self.step, site, int(np.prod(shape)), p).reshape(shape))

    def embed(self, n, L, H):
        return self._t(SITE_EMBED, (n, L, H), self.p_hidden)

    def attn_out(self, layer, n, L, H):
        return self._t(site_attn_out(layer), (n, L, H), self.p_hidden)

    def ffn_out(self, layer, n, L, H):
        return self._t(site_ffn_out(layer), (n, L, H), self.p_hidden)

    def probs(self, layer, n, A, L):
        import torch
        return torch.from_numpy(multipliers8(self.seed, self.step, site_probs(layer), n * A * L * L, self.p_attn).reshape(n, A, L, L))
