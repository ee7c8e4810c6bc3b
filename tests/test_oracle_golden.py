"""CPU: the oracles (oracle/torch_ref.py, oracle/qst_oracle.py) against the committed golden vectors that
oracle/make_golden.py produced from the REAL reference loss module and HF BertModel/MPNetModel."""
import os

import numpy as np
import pytest
import torch

import quadruplet_sentence_transformer_amd  # noqa: F401
from quadruplet_sentence_transformer_amd.config import PRESETS, build_layout
from quadruplet_sentence_transformer_amd.synthetic import approx_normal, synthetic_params, synthetic_quadruplets
from oracle import qst_oracle as NO
from oracle import torch_ref as R

CLI = dict(gamma=0.6, margin_pos_neg=1.0, margin_pos_part=0.5, margin_part_neg=0.5)
CLS = dict(gamma=0.6, margin_pos_neg=1.0, margin_pos_part=1.0, margin_part_neg=1.0)


@pytest.fixture(scope="module")
def loss_g(golden_dir):
    return np.load(os.path.join(golden_dir, "loss_golden.npz"))


@pytest.fixture(scope="module")
def enc_g(golden_dir):
    return np.load(os.path.join(golden_dir, "encoder_golden.npz"))


def loss_inputs(B, D, seed):
    x = approx_normal(seed, 1, 4 * B * D, 1.0).reshape(4, B, D)
    if D >= 384:
        x = x / np.linalg.norm(x, axis=-1, keepdims=True)
    return x.astype(np.float32)


LOSS_CASES = [(ci, B, D, p, swap, mk) for ci, (B, D) in enumerate([(1, 10), (5, 10), (8, 384), (8, 768), (32, 384)])
              for p in (2.0, 1.0) for swap in (False, True) for mk in ("cli", "cls") if not (mk == "cls" and (p != 2.0 or swap))]


@pytest.mark.parametrize("ci,B,D,p,swap,mk", LOSS_CASES)
def test_loss_oracles_match_reference(loss_g, ci, B, D, p, swap, mk):
    x = loss_inputs(B, D, 100 + ci)
    key = f"B{B}_D{D}_p{int(p)}_s{int(swap)}_{mk}"
    kw = dict(CLI if mk == "cli" else CLS, p=p, swap=swap)
    tol = dict(rtol=2e-5, atol=2e-5 if p == 2.0 else 2e-6 * D + 2e-5)
    # torch restatement, forward (3 reductions) + autograd backward
    t = [torch.from_numpy(x[i]).clone().requires_grad_(True) for i in range(4)]
    for red in ("none", "sum", "mean"):
        got = R.gamma_quadruplet_loss_ref(*t, reduction=red, **kw)
        np.testing.assert_allclose(got.detach().numpy(), loss_g[f"{key}_{red}"], **dict(tol, atol=tol["atol"] * (B if red == "sum" else 1)))
    R.gamma_quadruplet_loss_ref(*t, reduction="mean", **kw).backward()
    for i in range(4):
        np.testing.assert_allclose(t[i].grad.numpy(), loss_g[key + "_grads"][i], rtol=1e-4, atol=1e-6)
    # numpy restatement, forward + analytic backward
    for red in ("none", "sum", "mean"):
        got = NO.gamma_quadruplet_loss(*x.astype(np.float64), reduction=red, **kw)
        np.testing.assert_allclose(got, loss_g[f"{key}_{red}"], **dict(tol, atol=tol["atol"] * (B if red == "sum" else 1)))
    _, g = NO.gamma_quadruplet_loss(*x.astype(np.float64), reduction="mean", with_grads=True, **kw)
    for i in range(4):
        np.testing.assert_allclose(g[i], loss_g[key + "_grads"][i], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name", ["edge_inactive", "edge_all_equal", "edge_dup_rows"])
@pytest.mark.parametrize("swap", [False, True])
def test_loss_edge_cases(loss_g, name, swap):
    x = loss_g[f"{name}_s{int(swap)}_x"]
    t = [torch.from_numpy(x[i]).clone().requires_grad_(True) for i in range(4)]
    m = R.gamma_quadruplet_loss_ref(*t, swap=swap, **CLI)
    m.backward()
    np.testing.assert_allclose(m.item(), loss_g[f"{name}_s{int(swap)}_mean"], rtol=1e-5, atol=1e-6)
    for i in range(4):
        np.testing.assert_allclose(t[i].grad.numpy(), loss_g[f"{name}_s{int(swap)}_grads"][i], rtol=1e-4, atol=1e-6)
    l, g = NO.gamma_quadruplet_loss(*x.astype(np.float64), swap=swap, with_grads=True, **CLI)
    np.testing.assert_allclose(l, loss_g[f"{name}_s{int(swap)}_mean"], rtol=1e-5, atol=1e-6)
    for i in range(4):
        np.testing.assert_allclose(g[i], loss_g[f"{name}_s{int(swap)}_grads"][i], rtol=1e-4, atol=1e-6)


def test_loss_invariants_from_reference_notebook():
    # quadruplet_loss_test.ipynb cells 9/13: 'none' -> (B,), 'sum'/'mean' -> 0-d, sum/B == mean == none.mean()
    x = [torch.randn(5, 10, generator=torch.Generator().manual_seed(i)) for i in range(4)]
    none = R.gamma_quadruplet_loss_ref(*x, reduction="none")
    s = R.gamma_quadruplet_loss_ref(*x, reduction="sum")
    m = R.gamma_quadruplet_loss_ref(*x, reduction="mean")
    assert none.shape == (5,) and s.dim() == 0 and m.dim() == 0
    torch.testing.assert_close(s / 5, m)
    torch.testing.assert_close(none.mean(), m)


ENC_CASES = [("tinybert_hfinit", "tiny-bert", 2, 32, dict(std=0.02), "full"),
             ("tinybert_trained", "tiny-bert", 3, 64, dict(std=0.08, bias_std=0.05, ln_jitter=0.1), "full"),
             ("tinympnet_trained", "tiny-mpnet", 2, 64, dict(std=0.08, bias_std=0.05, ln_jitter=0.1), "full"),
             ("minilm_c1", "all-MiniLM-L6-v2", 8, 32, dict(std=0.04, bias_std=0.02, ln_jitter=0.05), "norms"),
             # an all-padding sequence and left-padded ones (synthetic.mask_edge_cases)
             ("tinybert_maskedge", "tiny-bert", 3, 64, dict(std=0.08, bias_std=0.05, ln_jitter=0.1), "full"),
             ("tinympnet_maskedge", "tiny-mpnet", 3, 64, dict(std=0.08, bias_std=0.05, ln_jitter=0.1), "full"),
             # 16,384 token rows: the size the fused GEMM + LayerNorm kernels, the 8-range wgrad and the single-workgroup
             # attention backward run from (the HIP side of this case: tests/test_gpu_golden.py)
             ("minilm2l_fused", "minilm-2l", 32, 128, dict(std=0.04, bias_std=0.02, ln_jitter=0.05), "norms"),
             # full dims of BASELINE configs[2] / configs[4], trained-like weights (round 5)
             ("mpnetbase_trained", "all-mpnet-base-v2", 1, 64, dict(std=0.04, bias_std=0.02, ln_jitter=0.05), "norms"),
             ("bertbase_trained", "bert-base-uncased", 1, 64, dict(std=0.04, bias_std=0.02, ln_jitter=0.05), "norms")]


def golden_inputs(key, cfg, B, L):
    """The inputs oracle/make_golden.py used for encoder case `key`."""
    from quadruplet_sentence_transformer_amd.synthetic import mask_edge_cases
    ids, mask, types = synthetic_quadruplets(cfg, B, L, seed=14, ragged=True)
    if key.endswith("maskedge"):
        ids, mask = mask_edge_cases(ids, mask, cfg.pad_token_id)
    return ids, mask, types


@pytest.mark.parametrize("key,preset,B,L,wkw,store", ENC_CASES)
def test_encoder_oracles_match_hf(enc_g, key, preset, B, L, wkw, store):
    cfg = PRESETS[preset]
    arena = synthetic_params(cfg, seed=14, **wkw)
    ids, mask, types = golden_inputs(key, cfg, B, L)
    P = R.arena_to_dict(arena, cfg, requires_grad=True)
    loss, emb = R.quadruplet_step(P, cfg, torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(types), CLI)
    np.testing.assert_allclose(emb.detach().numpy(), enc_g[key + "_emb"], rtol=1e-4, atol=2e-6)
    assert abs(loss.item() - float(enc_g[key + "_loss"])) < 2e-6
    loss.backward()
    segs, total = build_layout(cfg)
    g = np.zeros(total, np.float32)
    for s in segs:
        g[s.offset:s.offset + s.numel] = P[s.name].grad.numpy().reshape(-1)
    if store == "full":
        ref = enc_g[key + "_grads"]
        for s in segs:
            a, b = g[s.offset:s.offset + s.numel], ref[s.offset:s.offset + s.numel]
            assert np.linalg.norm(a - b) <= 2e-4 * np.linalg.norm(b) + 1e-7, s.name
    else:
        norms = np.array([np.linalg.norm(g[s.offset:s.offset + s.numel]) for s in segs])
        np.testing.assert_allclose(norms, enc_g[key + "_gradnorms"], rtol=2e-3, atol=1e-7)
    if B * L > 2048:
        return                                        # (the numpy restatement is for the small cases)
    # numpy restatement of the forward
    l2, e2 = NO.quadruplet_forward(NO.arena_to_dict(arena, cfg), cfg, ids, mask, types, **CLI)
    np.testing.assert_allclose(e2, enc_g[key + "_emb"], rtol=1e-3, atol=2e-5)
    assert abs(float(l2) - float(enc_g[key + "_loss"])) < 2e-5


def test_mpnet_bucket_tables_agree(enc_g):
    lut = enc_g["mpnet_bucket_lut"]
    rel = np.arange(-511, 512)
    np.testing.assert_array_equal(NO.mpnet_bucket(rel), lut)
    t = R.mpnet_bucket_table(512)
    np.testing.assert_array_equal(t[0].numpy(), lut[511:511 + 512])       # i = 0: rel = j
