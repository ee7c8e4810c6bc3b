"""End-to-end GPU parity: HIP encoder forward/backward + loss vs the CPU oracle (oracle/torch_ref.py).

Tolerances (DESIGN.md "Precision"): the bf16 path rounds every MFMA operand once to bf16.
  * vs the oracle run with the SAME operand rounding (bf16_operands=True): embeddings
    rtol 1e-3 / atol 1e-4 (BASELINE.json north_star tolerance), loss 1e-4.
  * vs the fp32 oracle: loss within 1e-3 (north_star target); embeddings atol 2e-3 (measured
    bf16 rounding error, not a kernel property).
  * gradients: relative L2 error per parameter tensor < 2e-2 vs autograd through the
    bf16-operand oracle.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import quadruplet_sentence_transformer_amd as qst  # noqa: E402
from quadruplet_sentence_transformer_amd.config import PRESETS, build_layout  # noqa: E402
from quadruplet_sentence_transformer_amd.encoder import HipEncoder, quadruplet_loss_raw  # noqa: E402
from quadruplet_sentence_transformer_amd.synthetic import synthetic_params, synthetic_quadruplets  # noqa: E402
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402
from oracle import torch_ref as R  # noqa: E402

# Gradient bounds (relative L2 error per parameter tensor) against the oracle that rounds the same GEMM operands to bf16 --
# in the forward AND in the backward (dY, dO, P, dS enter its products bf16-rounded, as they enter the matrix cores; bias
# gradients are column sums of the rounded dY) -- and accumulates the products in fp64 (oracle/torch_ref.py: independent of
# the box's BLAS reduction order). Each bound is the maximum measured over every case of this file and of
# test_gpu_dropout.py x 1.25 ([measured] in brackets). What is left between the two:
#   w      weight matrices, rel_bias -- accumulation order, and the saved activations the backward kernels read as bf16
#          (gelu'(u), the normalised rows) where autograd keeps fp32; largest on the smallest batch through all six layers;
#   emb    embedding tables -- sums of the gradient that has crossed every layer, over few rows per table row;
#   vec    bias / LayerNorm vectors -- sums over M rows of values that differ in their last bf16 bit between the two
#          implementations, largest on the smallest batch (M = 256 rows);
#   b_qkv  its key third has a mathematically zero gradient (softmax shift invariance), so a third of the vector is pure
#          rounding noise in both implementations.
# For scale: the arithmetic tests/test_gpu_f16.py derives its bound from -- 12 roundings per layer x 6 layers of independent relative
# errors of one unit roundoff, 12 x 6 x u / sqrt(12 x 6) -- gives 1.66e-2 for bf16 (u = 2^-9): the calibrated "w" bound is that figure,
# i.e. against the same-rounding oracle the kernels are allowed what ONE more set of bf16 roundings would cost (values on a rounding
# boundary that flip between the two implementations), no more.
GRAD_LIMITS = {"w": 1.65e-2, "emb": 1.55e-2, "vec": 2.2e-2, "b_qkv": 3.55e-2}     # [1.31e-2, 1.22e-2, 1.74e-2, 2.83e-2]

LOSS_KW = dict(gamma=0.6, margin_pos_neg=1.0, margin_pos_part=0.5, margin_part_neg=0.5, p=2.0, swap=False)


def cls_of(leaf):
    """Class of a parameter tensor (by the last component of its name) for the gradient bounds."""
    return ("b_qkv" if leaf == "b_qkv" else "vec" if (leaf.startswith("b_") or leaf.startswith("ln") or leaf.startswith("emb_ln"))
            else "emb" if leaf.endswith("_emb") else "w")


def run_case(name, B, L, ragged, weights_kw, check_grads=True, emb_atol_vs_bf16_oracle=1e-4, scale_by_emb=False,
             mask_edges=False, dropout=None, ffn_chain=None, ln_fusion=None):
    """dropout = (p_hidden, p_attn, seed): the HIP encoder runs its training forward / backward with dropout on, the
    oracle with the SAME masks (oracle/dropout_ref.py regenerates them from seed, step 1) -- same comparisons, same
    bounds."""
    cfg = PRESETS[name]
    arena = synthetic_params(cfg, seed=14, **weights_kw)
    ids, mask, types = synthetic_quadruplets(cfg, B, L, seed=14, ragged=ragged)
    if mask_edges:
        from quadruplet_sentence_transformer_amd.synthetic import mask_edge_cases
        ids, mask = mask_edge_cases(ids, mask, cfg.pad_token_id)
    ids_t, mask_t, types_t = torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(types)

    # oracle
    masks = None
    if dropout is not None:
        from oracle.dropout_ref import Masks
        masks = Masks(dropout[2], 1, dropout[0], dropout[1])
    P32 = R.arena_to_dict(arena, cfg)
    with torch.no_grad():
        loss32, emb32 = R.quadruplet_step(P32, cfg, ids_t, mask_t, types_t, LOSS_KW, dropout=masks)
    Pb = R.arena_to_dict(arena, cfg, requires_grad=check_grads)
    lossb, embb = R.quadruplet_step(Pb, cfg, ids_t, mask_t, types_t, LOSS_KW, bf16_operands=True, dropout=masks)
    if check_grads:
        lossb.backward()

    # HIP
    enc = HipEncoder(cfg)
    enc.load_arena(arena)
    if dropout is not None:
        enc.set_dropout(dropout[0], dropout[1], dropout[2])
    if ffn_chain is not None:
        enc.set_ffn_chain(ffn_chain)
    if ln_fusion is not None:
        enc.set_ln_fusion(ln_fusion)
    n = 4 * B
    idd, mdd, tdd = ids_t.view(n, L).cuda(), mask_t.view(n, L).cuda(), types_t.view(n, L).cuda()
    emb, tok, saved = enc.forward(idd, mdd, tdd if cfg.type_vocab_size else None, training=True, want_tokens=True)
    e4 = emb.view(4, B, -1)
    loss, g = quadruplet_loss_raw(e4[0], e4[1], e4[2], e4[3], 0.6, 1.0, 0.5, 0.5, 2.0, False, 2, want_grads=True)
    torch.cuda.synchronize()
    assert torch.isfinite(emb).all()
    # models without the Normalize module (bare bert-base) emit un-normalised embeddings: scale the absolute
    # tolerances by the embedding magnitude so they mean the same thing as for unit-norm outputs
    sc = float(emb32.norm(dim=-1).mean()) if scale_by_emb else 1.0
    torch.testing.assert_close(emb.cpu().view(4, B, -1), embb.detach(), rtol=1e-3, atol=emb_atol_vs_bf16_oracle * sc)
    assert abs(loss.item() - lossb.item()) < max(1e-4, 0.3 * emb_atol_vs_bf16_oracle) * sc
    assert abs(loss.item() - loss32.item()) < 1e-3 * sc
    torch.testing.assert_close(emb.cpu().view(4, B, -1), emb32, rtol=0, atol=2e-3 * sc)

    if check_grads:
        enc.ensure_train_state()
        enc.grads.zero_()
        enc.backward(idd, mdd, tdd if cfg.type_vocab_size else None, torch.cat(g, 0), saved)
        torch.cuda.synchronize()
        segs, _ = build_layout(cfg)
        ga = enc.grads.cpu()
        worst = 0.0
        errs = []
        cls_max = {}
        gnorm = float(torch.sqrt(sum((Pb[s.name].grad.double() ** 2).sum() for s in segs)))

        # a tensor whose gradient nearly cancels (bare bert-base's LAST feed-forward bias: 1/20 .. 1/40 of the other biases'
        # norm, tools/fuzz_shapes.py case 29) carries the same absolute rounding noise as its peers: it is measured against at
        # least 5% of the largest gradient norm of its class
        cls_top = {}
        for s in segs:
            c = cls_of(s.name.split(".")[-1])
            cls_top[c] = max(cls_top.get(c, 0.0), Pb[s.name].grad.norm().item())
        for s in segs:
            ref = Pb[s.name].grad
            got = ga[s.offset:s.offset + s.numel].view(*s.shape)
            denom = ref.norm().item()
            if denom <= 1e-5 * gnorm:         # (gnorm = 0: no hinge of the batch is active, every gradient is exactly zero)
                # a mathematically zero gradient -- e.g. the last LayerNorm's beta of a model WITHOUT the Normalize module (bare
                # bert-base): it shifts every embedding alike and the loss sees differences only -- is rounding noise on both sides
                assert got.norm().item() <= 1e-4 * gnorm, s.name
                continue
            cls = cls_of(s.name.split(".")[-1])
            err = ((got - ref).norm() / max(denom, 0.05 * cls_top[cls])).item()
            worst = max(worst, err)
            cls_max[cls] = max(cls_max.get(cls, 0.0), err)
            assert err < GRAD_LIMITS[cls], f"{name} grad {s.name}: relative L2 error {err:.3e} (ref norm {denom:.3e})"
            errs.append((err, s.name))
        print(f"[grad-cls] {name} B={B} L={L}: " + ", ".join(f"{k} {v:.2e}" for k, v in sorted(cls_max.items())))
        errs.sort(reverse=True)
        print(f"[grad-err] {name} B={B} L={L}: " + ", ".join(f"{n} {e:.2e}" for e, n in errs[:4]))
        return loss.item(), worst
    return loss.item(), None


@pytest.mark.parametrize("name,B,L,ragged", [("tiny-bert", 2, 32, False), ("tiny-bert", 3, 64, True),
                                             ("tiny-mpnet", 2, 32, True), ("tiny-mpnet", 2, 64, True)])
def test_tiny_models_hf_init(name, B, L, ragged):
    run_case(name, B, L, ragged, dict(std=0.02))


@pytest.mark.parametrize("name", ["tiny-bert", "tiny-mpnet"])
def test_tiny_models_trained_like(name):
    # larger weights, non-zero biases, perturbed LayerNorm: every term of the network matters
    # with large weights a value that sits on a bf16 rounding boundary flips between the two
    # implementations, so even the same-rounding oracle only agrees to the bf16 noise floor
    run_case(name, 3, 64, True, dict(std=0.08, bias_std=0.05, ln_jitter=0.1), emb_atol_vs_bf16_oracle=1.5e-3)


def test_minilm_dims_at_the_fused_layernorm_size():
    """M = 4*32*128 = 16384 token rows: from this size on the H = 384 forward/backward run the GEMM kernels with the
    LayerNorm (forward and backward) fused into their epilogues; smaller batches take the unfused pair. Two MiniLM
    layers and a small vocabulary keep the CPU oracle to a few seconds."""
    run_case("minilm-2l", 32, 128, True, dict(std=0.03, bias_std=0.02, ln_jitter=0.05), emb_atol_vs_bf16_oracle=1.5e-3)


@pytest.mark.parametrize("base,L", [("all-MiniLM-L6-v2", 128), ("all-MiniLM-L6-v2", 256), ("all-mpnet-base-v2", 256)])
def test_all_padding_and_left_padded_sequences(base, L):
    """An all-zero attention_mask row and left-padded rows (a whole leading 32-key tile masked), forward AND backward,
    on every attention code path: d = 32 single-workgroup backward (L = 128), d = 32 two-kernel backward (L = 256), d = 64
    with the relative-position bias. HF's finfo.min mask makes such a row attend uniformly and ST's clamp(min=1e-9) gives
    a zero embedding; a -inf mask gave NaN here, and one NaN row reaches every weight through the wgrad GEMMs."""
    from dataclasses import replace
    PRESETS["edge-2l"] = replace(PRESETS[base], num_layers=2, vocab_size=4096)
    try:
        run_case("edge-2l", 3, L, True, dict(std=0.03, bias_std=0.02, ln_jitter=0.05), emb_atol_vs_bf16_oracle=1.5e-3,
                 mask_edges=True)
    finally:
        del PRESETS["edge-2l"]


def test_feed_forward_block_as_one_kernel():
    """csrc/ffn.hip inside the encoder (M = 16384 rows, MiniLM layer dims): (a) the inference forward takes it by default
    and must give the embeddings of the training forward (two-kernel feed-forward path) -- same rounding points;
    (b) with the training variants switched on (HipEncoder.set_ffn_chain(7): forward saving gelu'(u) and h, backward
    producing du + LayerNorm-1 backward) the whole oracle comparison of run_case, gradients included, must still hold."""
    cfg = PRESETS["minilm-2l"]
    arena = synthetic_params(cfg, seed=14, std=0.03, bias_std=0.02, ln_jitter=0.05)
    ids, mask, types = [torch.from_numpy(x).view(128, 128).cuda() for x in synthetic_quadruplets(cfg, 32, 128, seed=14, ragged=True)]
    enc = HipEncoder(cfg)
    enc.load_arena(arena)
    e_train = enc.forward(ids, mask, types, training=True)[0].clone()
    e_inf = enc.forward(ids, mask, types, training=False)[0].clone()
    enc.set_ffn_chain(0)
    e_inf0 = enc.forward(ids, mask, types, training=False)[0].clone()
    torch.testing.assert_close(e_inf, e_train, rtol=0, atol=2e-5)       # bf16 h re-rounds identically; fp32 sums reorder
    torch.testing.assert_close(e_inf, e_inf0, rtol=0, atol=2e-5)
    run_case("minilm-2l", 32, 128, True, dict(std=0.03, bias_std=0.02, ln_jitter=0.05), emb_atol_vs_bf16_oracle=1.5e-3,
             ffn_chain=7)


def test_minilm_full_dims_ragged():
    run_case("all-MiniLM-L6-v2", 2, 128, True, dict(std=0.02))


def test_minilm_config1_shape():
    # BASELINE.json configs[0] shape: L=32, B=8
    run_case("all-MiniLM-L6-v2", 8, 32, True, dict(std=0.04, bias_std=0.02, ln_jitter=0.05), emb_atol_vs_bf16_oracle=1.5e-3)


def test_mpnet_base_full_dims_config3_shape():
    # BASELINE.json configs[2] architecture (12 layers, d_head 64, relative position bias), short ragged batch
    run_case("all-mpnet-base-v2", 1, 256, True, dict(std=0.02), emb_atol_vs_bf16_oracle=1e-3)


def test_bert_base_dims_l384():
    # BASELINE.json configs[4] architecture, all 12 layers, forward (bf16 operands here; tests/test_gpu_fp8mx.py runs the same
    # dims on the fp8 matrix cores): 3 key chunks of 128. Gradients at this sequence length: the two-layer case below (the CPU
    # oracle's autograd over 12 layers x 1,536 tokens is minutes).
    run_case("bert-base-uncased", 1, 384, True, dict(std=0.02), check_grads=False, emb_atol_vs_bf16_oracle=1e-3,
             scale_by_emb=True)


@pytest.mark.parametrize("drop", [None, (0.1, 0.1, 9)], ids=["eval", "train_mode_dropout"])
def test_bert_base_two_layers_l384_with_gradients(drop):
    """configs[4]'s sequence length WITH gradients (VERDICT r03 'missing' 5): bert-base layer dims (H = 768, d = 64, I = 3072),
    two layers, L = 384 ragged -- three 128-key chunks in the attention forward, the two-kernel d = 64 backward over three
    query / key blocks, the K >= 768 GEMM tiles and the H = 768 LayerNorm row kernels -- every gradient tensor against the
    same-rounding oracle, with and without the reference's train()-mode dropout."""
    from dataclasses import replace
    PRESETS["bert-2l"] = replace(PRESETS["bert-base-uncased"], num_layers=2, vocab_size=4096)
    try:
        # (bert-base-uncased has no Normalize module: embeddings of norm ~sqrt(H), tolerances on that scale as in the 12-layer case)
        run_case("bert-2l", 1, 384, True, dict(std=0.03, bias_std=0.02, ln_jitter=0.05), emb_atol_vs_bf16_oracle=1.5e-3, dropout=drop,
                 scale_by_emb=True)
    finally:
        del PRESETS["bert-2l"]


@pytest.mark.parametrize("base,L,drop", [("bert-base-uncased", 384, None), ("bert-base-uncased", 384, (0.1, 0.1, 9)),
                                         ("all-mpnet-base-v2", 288, (0.2, 0.1, 7))], ids=["bert", "bert_dropout", "mpnet_dropout"])
def test_layernorm_fused_across_the_tiles_of_a_row_h768(base, L, drop):
    """H = 768: every projection + LayerNorm and dgrad + LayerNorm backward as ONE launch whose three workgroups per 256-row
    panel exchange the row statistics (csrc/gemm8.hip gemm_nt8_ln_kernel; taken by size from two tiles per CU, forced here
    with set_ln_fusion(1) on 1,536 / 1,152 token rows -- the second not a multiple of the panel height): (a) the whole
    oracle comparison of run_case, every gradient tensor included; (b) against the unfused pair (set_ln_fusion(2)) on the
    same inputs, bf16 and f16 operands: same arithmetic up to fp32 summation order and the 16-bit roundings it flips."""
    from dataclasses import replace
    PRESETS["h768-2l"] = replace(PRESETS[base], num_layers=2, vocab_size=4096)
    try:
        cfg = PRESETS["h768-2l"]
        run_case("h768-2l", 1, L, True, dict(std=0.03, bias_std=0.02, ln_jitter=0.05), emb_atol_vs_bf16_oracle=1.5e-3, dropout=drop,
                 scale_by_emb=(base == "bert-base-uncased"), ln_fusion=1)
        arena = synthetic_params(cfg, seed=15, std=0.03, bias_std=0.02, ln_jitter=0.05)
        ids, mask, types = [torch.from_numpy(x).view(4, L).cuda() for x in synthetic_quadruplets(cfg, 1, L, seed=15, ragged=True)]
        tt = types if cfg.type_vocab_size else None
        for prec in ("bf16", "f16"):
            out = {}
            for mode in (2, 1):
                enc = HipEncoder(cfg)
                enc.load_arena(arena)
                if drop is not None:
                    enc.set_dropout(*drop)
                enc.set_ln_fusion(mode)
                enc.ensure_train_state()
                emb, _, saved = enc.forward(ids, mask, tt, training=True, precision=prec)
                gen = torch.Generator().manual_seed(3)
                ge = torch.randn(emb.shape, generator=gen).cuda()
                enc.grads.zero_()
                enc.backward(ids, mask, tt, ge, saved, precision=prec)
                torch.cuda.synchronize()
                out[mode] = (emb.clone(), enc.grads.clone())
            # (fp32 sums in another order flip the 16-bit rounding of a few y / xhat elements: one ulp of 2^-8 / 2^-11 each,
            #  measured 1.2e-4 of the embedding scale with bf16 operands)
            et, gt = (4e-4, 1e-2) if prec == "bf16" else (1e-4, 3e-3)
            sc = out[2][0].abs().max().item()
            de = (out[1][0] - out[2][0]).abs().max().item()
            assert de <= et * sc, f"{prec}: embeddings of the fused and the unfused path differ by {de:.3e} (scale {sc:.3e})"
            gs = out[2][1].abs().max().item()
            d = (out[1][1] - out[2][1]).abs().max().item()
            print(f"[ln-fusion] {prec}: max|d emb| {de:.2e} of {sc:.2e}, max|d grad| {d:.2e} of {gs:.2e}")
            assert d <= gt * gs, f"{prec}: gradients of the fused and the unfused path differ by {d:.3e} (largest gradient {gs:.3e})"
        assert enc.lib.qst_gemm_nt8_ln_timeouts() == 0 and enc.lib.qst_gemm_nt8_ln_timeouts_f16() == 0
    finally:
        del PRESETS["h768-2l"]


def test_layernorm_fused_on_the_128x384_tile_in_the_encoder():
    """M = 32,768 token rows at H = 768 (configs[2]'s row count): the size rule takes the GEMM + LayerNorm launches on their
    128 x 384 tile (two tiles per CU; 256 x 256 would leave a round and a half). One bert-base layer, 256 sequences x 128
    tokens, dropout on: embeddings and every gradient of the fused path (set_ln_fusion(0): by size) against the GEMM +
    row-kernel pair (set_ln_fusion(2)) on the same inputs and masks."""
    from dataclasses import replace
    cfg = replace(PRESETS["bert-base-uncased"], num_layers=1, vocab_size=4096)
    L, nseq = 128, 256
    arena = synthetic_params(cfg, seed=16, std=0.03, bias_std=0.02, ln_jitter=0.05)
    ids, mask, types = [torch.from_numpy(x).view(nseq, L).cuda() for x in synthetic_quadruplets(cfg, nseq // 4, L, seed=16, ragged=True)]
    lib = _lib.load()
    assert lib.qst_gemm_nt_ln_block_rows_m(768, nseq * L) == 128 and lib.qst_gemm_nt_ln_block_rows_m(768, 196608) == 256
    out = {}
    for mode in (2, 0):
        enc = HipEncoder(cfg)
        enc.load_arena(arena)
        enc.set_dropout(0.1, 0.1, 21)
        enc.set_ln_fusion(mode)
        enc.ensure_train_state()
        emb, _, saved = enc.forward(ids, mask, types, training=True)
        ge = torch.randn(emb.shape, generator=torch.Generator().manual_seed(3)).cuda()
        enc.grads.zero_()
        enc.backward(ids, mask, types, ge, saved)
        torch.cuda.synchronize()
        out[mode] = (emb.clone(), enc.grads.clone())
        del enc
    sc, gs = out[2][0].abs().max().item(), out[2][1].abs().max().item()
    de, dg = (out[0][0] - out[2][0]).abs().max().item(), (out[0][1] - out[2][1]).abs().max().item()
    print(f"[ln-fusion 128x384] max|d emb| {de:.2e} of {sc:.2e}, max|d grad| {dg:.2e} of {gs:.2e}")
    assert de <= 4e-4 * sc and dg <= 1e-2 * gs
    assert lib.qst_gemm_nt8_ln_timeouts() == 0


@pytest.mark.parametrize("name,B,L,ragged,wkw,drop", [
    ("tiny-bert", 3, 64, True, dict(std=0.08, bias_std=0.05, ln_jitter=0.1), None),
    ("tiny-mpnet", 2, 64, True, dict(std=0.08, bias_std=0.05, ln_jitter=0.1), None),
    ("tiny-bert", 2, 32, False, dict(std=0.02), None),
    ("minilm-2l", 2, 128, True, dict(std=0.03, bias_std=0.02, ln_jitter=0.05), None),
    ("mpnet-2l", 1, 288, True, dict(std=0.03, bias_std=0.02, ln_jitter=0.05), None),      # d = 64, two key blocks (256 + 32)
    # train() mode, as the reference's fit() runs (fp32 + HF dropout 0.1 / 0.1): the oracle gets the same masks
    ("tiny-bert", 3, 64, True, dict(std=0.08, bias_std=0.05, ln_jitter=0.1), (0.1, 0.1, 5)),
    ("minilm-2l", 2, 128, True, dict(std=0.03, bias_std=0.02, ln_jitter=0.05), (0.1, 0.1, 6)),
    ("mpnet-2l", 1, 288, True, dict(std=0.03, bias_std=0.02, ln_jitter=0.05), (0.2, 0.1, 7))])
def test_parity_precision_backward_matches_fp32_autograd(name, B, L, ragged, wkw, drop):
    """precision="bf16x3" TRAINING (the reference trains in fp32, training/main.py:142): forward(training=True) + backward on
    the split-bf16 x3 path against fp32 torch autograd -- embeddings within the north-star atol 1e-4, loss 1e-5, every
    gradient tensor within 1e-4 relative L2 (measured 1.2e-5 ... 2.1e-5; the bf16 path's bounds are 1.65e-2 ... 3.55e-2). The loss gradient comes from
    the HIP loss kernel on the x3 embeddings. mpnet-2l: mpnet-base dims (d = 64, position bias) at L = 288 -- two key blocks in
    the fp32 attention backward."""
    from dataclasses import replace
    cfg = replace(PRESETS["all-mpnet-base-v2"], num_layers=2, vocab_size=4096) if name == "mpnet-2l" else PRESETS[name]
    arena = synthetic_params(cfg, seed=21, **wkw)
    ids, mask, types = synthetic_quadruplets(cfg, B, L, seed=21, ragged=ragged)
    ids_t, mask_t, types_t = [torch.from_numpy(x) for x in (ids, mask, types)]
    P = R.arena_to_dict(arena, cfg, requires_grad=True)
    masks = None
    if drop is not None:
        from oracle.dropout_ref import Masks
        masks = Masks(drop[2], 1, drop[0], drop[1])
    loss32, emb32 = R.quadruplet_step(P, cfg, ids_t, mask_t, types_t if cfg.type_vocab_size else None, LOSS_KW, bf16_operands=False,
                                      dropout=masks)
    loss32.backward()
    enc = HipEncoder(cfg)
    enc.load_arena(arena)
    enc.ensure_train_state()
    if drop is not None:
        enc.set_dropout(drop[0], drop[1], drop[2])
    n = 4 * B
    idd, mdd, tdd = ids_t.view(n, L).cuda(), mask_t.view(n, L).cuda(), types_t.view(n, L).cuda()
    tdd = tdd if cfg.type_vocab_size else None
    emb, _, saved = enc.forward(idd, mdd, tdd, training=True, precision="bf16x3")
    e4 = emb.view(4, B, -1)
    loss, g = quadruplet_loss_raw(e4[0], e4[1], e4[2], e4[3], 0.6, 1.0, 0.5, 0.5, 2.0, False, 2, want_grads=True)
    enc.grads.zero_()
    enc.backward(idd, mdd, tdd, torch.cat(g, 0), saved, precision="bf16x3")
    torch.cuda.synchronize()
    torch.testing.assert_close(emb.cpu().view(4, B, -1), emb32.detach(), rtol=1e-3, atol=1e-4)
    assert abs(loss.item() - loss32.item()) < 1e-5
    segs, _ = build_layout(cfg)
    ga = enc.grads.cpu()
    worst = (0.0, "")
    gnorm = float(torch.sqrt(sum((P[s_.name].grad.double() ** 2).sum() for s_ in segs)))
    cls_top = {}
    for s_ in segs:
        c = cls_of(s_.name.split(".")[-1])
        cls_top[c] = max(cls_top.get(c, 0.0), P[s_.name].grad.norm().item())
    for s_ in segs:
        ref = P[s_.name].grad
        got = ga[s_.offset:s_.offset + s_.numel].view(*s_.shape)
        denom = ref.norm().item()
        if denom <= 1e-5 * gnorm:                # a mathematically zero gradient (see run_case): rounding noise on both sides
            assert got.norm().item() <= 1e-5 * gnorm, s_.name
            continue
        # (near-cancelling tensors -- the last LayerNorm's beta, the last feed-forward bias -- on the scale of their class, as run_case)
        err = ((got - ref).norm() / max(denom, 0.05 * cls_top[cls_of(s_.name.split(".")[-1])])).item()
        worst = max(worst, (err, s_.name))
        assert err < 1e-4, f"{name} grad {s_.name}: relative L2 error {err:.3e} (ref norm {denom:.3e})"     # measured <= 2.1e-5
    print(f"[x3-grad-err] {name} B={B} L={L}: worst {worst[1]} {worst[0]:.2e}")
