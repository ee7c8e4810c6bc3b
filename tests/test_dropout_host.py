"""CPU checks of the dropout mask generator as restated in oracle/dropout_ref.py (the GPU tests prove the kernels produce
exactly these masks): it must behave like torch.nn.Dropout's Bernoulli(1 - p) masks in distribution -- the reference's
train() mode (HF hidden_dropout_prob = attention_probs_dropout_prob = 0.1) fixes nothing else."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from oracle import dropout_ref as D  # noqa: E402
from oracle import torch_ref as R  # noqa: E402
from quadruplet_sentence_transformer_amd.config import PRESETS  # noqa: E402
from quadruplet_sentence_transformer_amd.synthetic import synthetic_params, synthetic_quadruplets  # noqa: E402


def test_threshold_and_scale():
    assert D.thr16_of(0.1) == 6554 and D.thr16_of(0.0) == 0 and D.thr16_of(0.5) == 32768 and D.thr16_of(0.99999) == 65535
    m = D.multipliers(7, 1, 3, 4096, 0.25)
    assert set(np.unique(m)) == {np.float32(0.0), np.float32(65536.0 / (65536 - 16384))}
    assert np.all(D.multipliers(7, 1, 3, 100, 0.0) == 1.0)


def test_rate_and_unbiasedness():
    n = 1 << 20
    for p in (0.1, 0.3):
        m = D.multipliers(123456789012345, 17, D.site_probs(3), n, p)
        rate = float((m == 0).mean())
        assert abs(rate - p) < 4 * np.sqrt(p * (1 - p) / n) + 1e-4, rate         # 4 sigma + the 2^-16 quantisation of p
        assert abs(float(m.mean()) - 1.0) < 3e-3                                 # E[mask] = 1


def test_eight_bit_form_of_the_attention_masks():
    assert D.thr8_of(0.1) == 26 and D.thr8_of(0.5) == 128 and D.thr8_of(0.0) == 0
    n = 1 << 20
    m = D.multipliers8(99, 5, D.site_probs(0), n, 0.1)
    assert set(np.unique(m)) == {np.float32(0.0), np.float32(256.0 / 230.0)}
    rate = float((m == 0).mean())
    assert abs(rate - 26 / 256) < 4 * np.sqrt(0.1 * 0.9 / n)
    assert abs(float(m.mean()) - 1.0) < 3e-3
    z = m == 0
    for k in (1, 2, 3):                                       # the four bytes of one word are independent of each other
        assert abs(float((z[0::4] & z[k::4]).mean()) - (26 / 256) ** 2) < 2e-3
    assert abs(float((z[:-4] & z[4:]).mean()) - (26 / 256) ** 2) < 1.5e-3
    other = D.multipliers8(99, 6, D.site_probs(0), n, 0.1) == 0
    assert abs(float((z & other).mean()) - (26 / 256) ** 2) < 1.5e-3


def test_streams_are_independent():
    n = 1 << 18
    base = D.multipliers(5, 1, D.site_attn_out(0), n, 0.1) == 0
    others = [D.multipliers(5, 2, D.site_attn_out(0), n, 0.1) == 0,            # next step
              D.multipliers(5, 1, D.site_ffn_out(0), n, 0.1) == 0,             # other tensor
              D.multipliers(5, 1, D.site_attn_out(1), n, 0.1) == 0,            # other layer
              D.multipliers(6, 1, D.site_attn_out(0), n, 0.1) == 0,            # other seed
              D.multipliers(5 + (1 << 32), 1, D.site_attn_out(0), n, 0.1) == 0]  # seed differing in the high word only
    for o in others:
        joint = float((base & o).mean())
        assert abs(joint - 0.01) < 1.5e-3, joint                                  # p^2 for independent masks
    # the two halves of one random word (elements 2i, 2i+1) and neighbouring words
    assert abs(float((base[0::2] & base[1::2]).mean()) - 0.01) < 2e-3
    assert abs(float((base[:-2] & base[2:]).mean()) - 0.01) < 1.5e-3
    # same arguments, same mask
    assert np.array_equal(base, D.multipliers(5, 1, D.site_attn_out(0), n, 0.1) == 0)


def test_oracle_with_zero_probability_is_eval_mode():
    cfg = PRESETS["tiny-bert"]
    P = R.arena_to_dict(synthetic_params(cfg, seed=3), cfg)
    ids, mask, types = [torch.from_numpy(x) for x in synthetic_quadruplets(cfg, 2, 32, seed=3, ragged=True)]
    with torch.no_grad():
        l0, e0 = R.quadruplet_step(P, cfg, ids, mask, types)
        l1, e1 = R.quadruplet_step(P, cfg, ids, mask, types, dropout=D.Masks(1, 1, 0.0, 0.0))
        l2, e2 = R.quadruplet_step(P, cfg, ids, mask, types, dropout=D.Masks(1, 1, 0.1, 0.1))
    assert torch.equal(e0, e1) and torch.equal(l0, l1)
    assert (e0 - e2).abs().max() > 1e-3                      # and a real mask changes the embeddings
