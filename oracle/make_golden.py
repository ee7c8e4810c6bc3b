#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference. Runs only in the build container:

    python oracle/make_golden.py

  * loss vectors come from /root/reference/models/losses/losses.py (imported, torch-only) --
    the reference's own GammaQuadrupletLoss / gamma_quadruplet_loss, forward and autograd backward;
  * encoder vectors come from transformers' BertModel / MPNetModel built from a config object
    (attn_implementation="eager", eval, fp32) + the restated ST head (mean pool clamp 1e-9, F.normalize) +
    the reference loss on top. The reference's glue/driver modules are not importable offline
    (sentence_transformers / nltk downloads) and are not needed for the numbers.

Weights and inputs are NOT stored: they are regenerated bit-identically from
quadruplet_sentence_transformer_amd.synthetic (integer hashing only). The .npz files hold expected outputs.
Nothing here travels to the GPU box except the fixtures it writes.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from models.losses.losses import GammaQuadrupletLoss, gamma_quadruplet_loss  # noqa: E402  (the reference)
from transformers import BertConfig, BertModel, MPNetConfig, MPNetModel  # noqa: E402

from quadruplet_sentence_transformer_amd.config import PRESETS, build_layout, hf_param_views  # noqa: E402
from quadruplet_sentence_transformer_amd.synthetic import (approx_normal, mask_edge_cases,  # noqa: E402
                                                           synthetic_params, synthetic_quadruplets)

OUT = os.path.join(ROOT, "tests", "golden")
CLI = dict(gamma=0.6, margin_pos_neg=1.0, margin_pos_part=0.5, margin_part_neg=0.5)   # training/main.py:211-218


def loss_inputs(B, D, seed):
    """[4, B, D] fp32, regenerated identically by the tests."""
    x = approx_normal(seed, 1, 4 * B * D, 1.0).reshape(4, B, D)
    if D >= 384:                      # sentence-embedding-like: unit rows
        x = x / np.linalg.norm(x, axis=-1, keepdims=True)
    return x.astype(np.float32)


def gen_loss():
    out = {}
    cases = [(1, 10), (5, 10), (8, 384), (8, 768), (32, 384)]
    for ci, (B, D) in enumerate(cases):
        x = loss_inputs(B, D, 100 + ci)
        for p in (2.0, 1.0):
            for swap in (False, True):
                for margins, mk in ((CLI, "cli"), (dict(gamma=0.6, margin_pos_neg=1.0, margin_pos_part=1.0,
                                                         margin_part_neg=1.0), "cls")):
                    if mk == "cls" and (p != 2.0 or swap):
                        continue
                    key = f"B{B}_D{D}_p{int(p)}_s{int(swap)}_{mk}"
                    t = [torch.from_numpy(x[i]).clone().requires_grad_(True) for i in range(4)]
                    none = gamma_quadruplet_loss(*t, p=p, swap=swap, reduction="none", **margins)
                    s = gamma_quadruplet_loss(*t, p=p, swap=swap, reduction="sum", **margins)
                    m = gamma_quadruplet_loss(*t, p=p, swap=swap, reduction="mean", **margins)
                    m.backward()
                    out[key + "_none"] = none.detach().numpy()
                    out[key + "_sum"] = s.detach().numpy()
                    out[key + "_mean"] = m.detach().numpy()
                    out[key + "_grads"] = np.stack([ti.grad.numpy() for ti in t])
                    # class wrapper == function, per-call reduction override (quadruplet_loss_test.ipynb cell 13)
                    mod = GammaQuadrupletLoss(p=p, swap=swap, reduction="sum", **margins)
                    assert torch.equal(mod(*[ti.detach() for ti in t]), s.detach())
                    assert torch.equal(mod(*[ti.detach() for ti in t], reduction="none"), none.detach())
    # edge cases
    B, D = 6, 64
    a = approx_normal(7, 3, B * D, 1.0).reshape(B, D)
    edge = {
        "edge_inactive": np.stack([a, a + 1e-3, a + 2e-3, a + 100.0]),          # pos/neg and part/neg hinges off
        "edge_all_equal": np.stack([a, a, a, a]),                                  # every distance = ||1e-6 * 1||
        "edge_dup_rows": np.stack([np.repeat(a[:1], B, 0), np.repeat(a[1:2], B, 0), np.repeat(a[2:3], B, 0),
                                   np.repeat(a[3:4], B, 0)]),
    }
    for k, x in edge.items():
        x = x.astype(np.float32)
        for swap in (False, True):
            t = [torch.from_numpy(x[i]).clone().requires_grad_(True) for i in range(4)]
            m = gamma_quadruplet_loss(*t, swap=swap, reduction="mean", **CLI)
            m.backward()
            out[f"{k}_s{int(swap)}_x"] = x
            out[f"{k}_s{int(swap)}_mean"] = m.detach().numpy()
            out[f"{k}_s{int(swap)}_none"] = gamma_quadruplet_loss(*[ti.detach() for ti in t], swap=swap,
                                                                  reduction="none", **CLI).numpy()
            out[f"{k}_s{int(swap)}_grads"] = np.stack([ti.grad.numpy() for ti in t])
    np.savez_compressed(os.path.join(OUT, "loss_golden.npz"), **out)
    print("loss_golden.npz:", len(out), "arrays")


def hf_model(cfg, arena):
    if cfg.arch == 0:
        c = BertConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_layers,
                       num_attention_heads=cfg.num_heads, intermediate_size=cfg.intermediate_size,
                       max_position_embeddings=cfg.max_position, type_vocab_size=cfg.type_vocab_size,
                       layer_norm_eps=cfg.layer_norm_eps, hidden_act="gelu")
        c._attn_implementation = "eager"
        m = BertModel(c)
    else:
        c = MPNetConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_layers,
                        num_attention_heads=cfg.num_heads, intermediate_size=cfg.intermediate_size,
                        max_position_embeddings=cfg.max_position, layer_norm_eps=cfg.layer_norm_eps,
                        relative_attention_num_buckets=cfg.rel_buckets, hidden_act="gelu")
        c._attn_implementation = "eager"
        m = MPNetModel(c)
    m.eval()
    segs, _ = build_layout(cfg)
    so = {s.name: s for s in segs}
    sd = dict(m.named_parameters())
    with torch.no_grad():
        for name, seg, off, shape in hf_param_views(cfg):
            s = so[seg]
            n = int(np.prod(shape))
            sd[name].copy_(torch.from_numpy(arena[s.offset + off:s.offset + off + n].reshape(shape)))
    return m


def hf_step(cfg, arena, ids, mask, types, loss_kw):
    """Real HF encoder + restated ST head + REAL reference loss; returns loss, emb, tok, arena-shaped grads."""
    m = hf_model(cfg, arena)
    four, B, L = ids.shape
    ids_t = torch.from_numpy(ids).view(4 * B, L)
    mask_t = torch.from_numpy(mask).view(4 * B, L)
    kw = dict(input_ids=ids_t, attention_mask=mask_t, return_dict=False)
    if cfg.arch == 0:
        kw["token_type_ids"] = torch.from_numpy(types).view(4 * B, L)
    tok = m(**kw)[0]
    mm = mask_t[:, :, None].float()
    emb = (tok * mm).sum(1) / mm.sum(1).clamp(min=1e-9)            # ST Pooling(mean)
    if cfg.normalize:
        emb = torch.nn.functional.normalize(emb, p=2, dim=1)        # ST Normalize
    e4 = emb.view(4, B, -1)
    loss = gamma_quadruplet_loss(e4[0], e4[1], e4[2], e4[3], **loss_kw)
    loss.backward()
    segs, total = build_layout(cfg)
    so = {s.name: s for s in segs}
    g = np.zeros(total, np.float32)
    sd = dict(m.named_parameters())
    for name, seg, off, shape in hf_param_views(cfg):
        s = so[seg]
        n = int(np.prod(shape))
        gr = sd[name].grad
        g[s.offset + off:s.offset + off + n] = (gr.numpy().reshape(-1) if gr is not None else 0.0)
    return loss.item(), e4.detach().numpy(), tok.detach().numpy(), g


def gen_encoder():
    out = {}
    cases = [
        # key, preset, B, L, ragged, weights kw, store
        ("tinybert_hfinit", "tiny-bert", 2, 32, True, dict(std=0.02), "full"),
        ("tinybert_trained", "tiny-bert", 3, 64, True, dict(std=0.08, bias_std=0.05, ln_jitter=0.1), "full"),
        ("tinympnet_trained", "tiny-mpnet", 2, 64, True, dict(std=0.08, bias_std=0.05, ln_jitter=0.1), "full"),
        ("minilm_c1", "all-MiniLM-L6-v2", 8, 32, True, dict(std=0.04, bias_std=0.02, ln_jitter=0.05), "norms"),
        ("minilm_l128", "all-MiniLM-L6-v2", 2, 128, True, dict(std=0.02), "norms"),
        # an all-padding sequence and two left-padded ones (synthetic.mask_edge_cases): HF's finfo.min mask + ST's clamp
        ("tinybert_maskedge", "tiny-bert", 3, 64, "edge", dict(std=0.08, bias_std=0.05, ln_jitter=0.1), "full"),
        ("tinympnet_maskedge", "tiny-mpnet", 3, 64, "edge", dict(std=0.08, bias_std=0.05, ln_jitter=0.1), "full"),
        # M = 4 * 32 * 128 = 16,384 token rows: the size from which the step takes its fused GEMM + LayerNorm kernels, the
        # 8-range grouped wgrad and the single-workgroup attention backward (VERDICT r02 missing #5)
        ("minilm2l_fused", "minilm-2l", 32, 128, True, dict(std=0.04, bias_std=0.02, ln_jitter=0.05), "norms"),
        # full dims of BASELINE configs[2] / configs[4] (12 layers, H = 768, d = 64), trained-like weights, 4 x 64 tokens: the
        # depth at which operand rounding has accumulated the most (round 5: the cases the f16 precision is judged on too)
        ("mpnetbase_trained", "all-mpnet-base-v2", 1, 64, True, dict(std=0.04, bias_std=0.02, ln_jitter=0.05), "norms"),
        ("bertbase_trained", "bert-base-uncased", 1, 64, True, dict(std=0.04, bias_std=0.02, ln_jitter=0.05), "norms"),
    ]
    for key, preset, B, L, ragged, wkw, store in cases:
        cfg = PRESETS[preset]
        arena = synthetic_params(cfg, seed=14, **wkw)
        ids, mask, types = synthetic_quadruplets(cfg, B, L, seed=14, ragged=bool(ragged))
        if ragged == "edge":
            ids, mask = mask_edge_cases(ids, mask, cfg.pad_token_id)
        loss, emb, tok, g = hf_step(cfg, arena, ids, mask, types, CLI)
        out[key + "_loss"] = np.float32(loss)
        out[key + "_emb"] = emb.astype(np.float32)
        segs, _ = build_layout(cfg)
        if store == "full":
            out[key + "_tok"] = tok.astype(np.float32)
            out[key + "_grads"] = g
        else:
            out[key + "_gradnorms"] = np.array([np.linalg.norm(g[s.offset:s.offset + s.numel]) for s in segs], np.float32)
            out[key + "_gradslices"] = np.stack([np.resize(g[s.offset:s.offset + min(64, s.numel)], 64) for s in segs])
        print(key, "loss", loss, "emb mean|x|", np.abs(emb).mean())
    # MPNet bucket table as torch computes it (rel = j - i in [-511, 511])
    enc = MPNetModel(MPNetConfig(vocab_size=8, hidden_size=32, num_hidden_layers=1, num_attention_heads=2,
                                 intermediate_size=32, max_position_embeddings=16)).encoder
    rel = torch.arange(-511, 512)[None, :]
    out["mpnet_bucket_lut"] = enc.relative_position_bucket(rel, num_buckets=32, max_distance=128).numpy().reshape(-1)
    np.savez_compressed(os.path.join(OUT, "encoder_golden.npz"), **out)
    print("encoder_golden.npz:", len(out), "arrays")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    gen_loss()
    gen_encoder()
