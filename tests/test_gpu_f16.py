"""GPU: QST_PREC_F16 -- the bf16 path's kernels compiled on IEEE-half matrix-core operands (csrc/qst_common.h: op16) --
against the committed HF golden vectors, the f16-operand oracle, and torch's AdamW under GradScaler's rules.

What the reference has in this place: `use_amp=True` = torch.cuda.amp.autocast (fp16) + GradScaler inside
SentenceTransformer.fit (/root/reference/training/main.py:142, default False :203) and autocast around the validation
loss (/root/reference/models/evaluators.py:92-94).

Tolerances are stated, not calibrated: forward = the north-star tolerance itself (rtol 1e-3 / atol 1e-4 on embeddings, 1e-3 on
the loss) for every golden case of up to six layers; gradients <= 5e-3 relative L2 per tensor against fp32 autograd
(each operand carries 2^-12 relative rounding error, an entry of a gradient tensor is a sum over a dozen rounded products along
a path of <= 6 layers: 12 x 6 x 2^-12 / sqrt(12 x 6) ~ 2e-3 for independent errors; 5e-3 leaves 2.5x). The two 12-layer
full-dims cases with trained-like weights are OUTSIDE the north-star tolerance with f16 operands in the oracle already
(tools/f16_oracle_check.py: 1.7e-4 / 6.4e-3 against 2.7e-4 ... 1.6e-3 / 4.6e-2 for bf16) and are asserted against that."""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402
from quadruplet_sentence_transformer_amd.config import PRESETS, build_layout  # noqa: E402
from quadruplet_sentence_transformer_amd.encoder import HipEncoder, quadruplet_loss_raw, stacked  # noqa: E402
from quadruplet_sentence_transformer_amd.synthetic import synthetic_params, synthetic_quadruplets  # noqa: E402
from quadruplet_sentence_transformer_amd.trainer import QuadrupletTrainer, warmup_linear_lr  # noqa: E402
from oracle import torch_ref as R  # noqa: E402
from tests.test_oracle_golden import CLI, ENC_CASES, golden_inputs  # noqa: E402

LOSS_KW = dict(gamma=0.6, margin_pos_neg=1.0, margin_pos_part=0.5, margin_part_neg=0.5, p=2.0, swap=False)
ALL_CASES = ENC_CASES + [("minilm_l128", "all-MiniLM-L6-v2", 2, 128, dict(std=0.02), "norms")]
# Cases a precision is NOT held to the north-star tolerance on: (max |d emb|, |d loss|) bounds instead = what the f16-operand
# ORACLE itself has against the HF vectors (tools/f16_oracle_check.py; /tmp experiments recorded in DESIGN.md finding 34) x 1.5.
#   plain f16: the two 12-layer full-dims cases (oracle 1.7e-4 with 47 of 3,072 elements outside / 6.4e-3), and minilm_c1, which
#   sits ON the edge -- oracle 1.23e-4 (inside, by where the largest error fell), HIP 1.14e-4 with 2 of 12,288 elements outside.
#   f16w (split weights): only bert-base (no Normalize module: unnormalised embeddings of magnitude ~1 after 12 layers).
LOOSE = {"f16": {"mpnetbase_trained": (2.6e-4, 1e-3), "bertbase_trained": (1.0e-2, 5e-3), "minilm_c1": (1.5e-4, 1e-3)},
         "f16w": {"bertbase_trained": (3.0e-3, 1e-2)}}


@pytest.fixture(scope="module")
def enc_g(golden_dir):
    return np.load(os.path.join(golden_dir, "encoder_golden.npz"))


def run_f16(cfg, arena, ids, mask, types, B, L, want_grads=True, scale=None, prec="f16"):
    """One f16 training forward + loss (+ backward under the loss scale `scale`, gradients returned UNSCALED)."""
    enc = HipEncoder(cfg)
    enc.load_arena(arena)
    n = 4 * B
    idd = torch.from_numpy(ids).view(n, L).cuda()
    mdd = torch.from_numpy(mask).view(n, L).cuda()
    tdd = torch.from_numpy(types).view(n, L).cuda() if cfg.type_vocab_size else None
    emb, _, saved = enc.forward(idd, mdd, tdd, training=True, precision=prec)
    e4 = emb.view(4, B, -1)
    gout = None
    if scale is not None:
        gout = torch.tensor([float(scale)], dtype=torch.float32, device="cuda")
    loss, g = quadruplet_loss_raw(e4[0], e4[1], e4[2], e4[3], 0.6, 1.0, 0.5, 0.5, 2.0, False, 2, grad_out=gout,
                                  want_grads=want_grads)
    ga = None
    if want_grads:
        enc.ensure_train_state()
        enc.grads.zero_()
        enc.backward(idd, mdd, tdd, stacked(g), saved, precision=prec)
        ga = enc.grads.cpu().numpy() / (1.0 if scale is None else float(scale))
    return loss.item(), e4.cpu().numpy(), ga, enc


@pytest.mark.parametrize("prec", ["f16", "f16w"])
@pytest.mark.parametrize("key,preset,B,L,wkw,store", ALL_CASES)
def test_f16_encoder_matches_hf_vectors_at_the_north_star_tolerance(enc_g, key, preset, B, L, wkw, store, prec):
    """Embeddings rtol 1e-3 / atol 1e-4 (element by element) and loss within 1e-3 of the fp32 HF reference vectors
    (BASELINE.json north_star): precision "f16w" (split weights) on every golden case but unnormalised bert-base, plain "f16" on
    every case of up to six layers but minilm_c1, which it straddles (LOOSE above) -- the bf16 precision is outside on seven of
    the eight six-layer cases. Gradients of a backward under GradScaler's initial scale (65536) within 5e-3 relative L2 per
    tensor of fp32 autograd."""
    cfg = PRESETS[preset]
    arena = synthetic_params(cfg, seed=14, **wkw)
    ids, mask, types = golden_inputs(key, cfg, B, L)
    loss, e4, ga, _ = run_f16(cfg, arena, ids, mask, types, B, L, scale=65536.0, prec=prec)
    ref = enc_g[key + "_emb"]
    if key in LOOSE[prec]:
        eb, lb = LOOSE[prec][key]
        assert np.abs(e4 - ref).max() <= eb and abs(loss - float(enc_g[key + "_loss"])) <= lb
        if key == "minilm_c1":
            assert (np.abs(e4 - ref) > 1e-4 + 1e-3 * np.abs(ref)).sum() <= 4          # of 12,288
    else:
        np.testing.assert_allclose(e4, ref, rtol=1e-3, atol=1e-4)
        assert abs(loss - float(enc_g[key + "_loss"])) < 1e-3
    if key.endswith("maskedge"):
        assert (e4[0, 1] == 0).all()                     # the all-padding sequence: exactly HF + ST's zero embedding
    assert np.isfinite(ga).all()
    segs, _ = build_layout(cfg)
    deep = cfg.num_layers > 6
    lim = 5e-3 if not deep else 1.5e-2
    if store == "full":
        refg = enc_g[key + "_grads"]
        top = max(np.linalg.norm(refg[s.offset:s.offset + s.numel]) for s in segs)
        for s in segs:
            a, b = ga[s.offset:s.offset + s.numel], refg[s.offset:s.offset + s.numel]
            if np.linalg.norm(b) < 1e-6 * top:
                continue                                  # a mathematically zero gradient (the key bias): rounding noise in both
            # b_qkv: two thirds of it carry signal, the key third is rounding noise of a zero gradient in both
            bound = lim if not s.name.endswith("b_qkv") else 2 * lim
            assert np.linalg.norm(a - b) <= bound * np.linalg.norm(b) + 1e-7, (s.name, np.linalg.norm(a - b) / np.linalg.norm(b))
    else:
        norms = np.array([np.linalg.norm(ga[s.offset:s.offset + s.numel]) for s in segs])
        rn = enc_g[key + "_gradnorms"]
        # (segments whose gradient is zero by symmetry are left out: without a Normalize module the four embedding gradients of
        #  a quadruplet sum to zero, so the last LayerNorm's beta -- bert-base -- receives rounding noise only, in every precision)
        keep = rn > 1e-4 * np.median(rn)
        np.testing.assert_allclose(norms[keep], rn[keep], rtol=lim, atol=1e-7)
        for k, s in enumerate(segs):
            refs = enc_g[key + "_gradslices"][k][:min(64, s.numel)]
            got = ga[s.offset:s.offset + min(64, s.numel)]
            if keep[k] and np.linalg.norm(refs) > 1e-2 * max(1e-12, rn[k]) / math.sqrt(max(1, s.numel / 64)):
                assert np.linalg.norm(got - refs) <= 4 * lim * np.linalg.norm(refs) + 1e-7, \
                    (s.name, np.linalg.norm(got - refs) / np.linalg.norm(refs))


@pytest.mark.parametrize("name,B,L", [("tiny-bert", 3, 64), ("tiny-mpnet", 2, 64), ("minilm-2l", 2, 128)])
def test_f16_path_equals_the_f16_operand_oracle(name, B, L):
    """Same rounding points on both sides (oracle/torch_ref.py, bf16_operands="f16"): what is left is accumulation order, the
    erf / exp approximations and values that sit on an f16 rounding boundary and flip (each flip moves one operand by 2^-11
    relative, so the two sides end up about as far from each other as either is from the fp32 result: measured 1.5e-6 ... 8.9e-5
    on the golden cases, tools/f16_gpu_report.py). Embeddings at the north-star tolerance, gradients 2.5e-3 relative L2."""
    cfg = PRESETS[name]
    arena = synthetic_params(cfg, seed=14, std=0.05, bias_std=0.02, ln_jitter=0.05)
    ids, mask, types = synthetic_quadruplets(cfg, B, L, seed=14, ragged=True)
    S = 4096.0
    loss, e4, ga, _ = run_f16(cfg, arena, ids, mask, types, B, L, scale=S)
    P = R.arena_to_dict(arena, cfg, requires_grad=True)
    lo, eo = R.quadruplet_step(P, cfg, torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(types), CLI, bf16_operands="f16")
    np.testing.assert_allclose(e4, eo.detach().numpy(), rtol=1e-3, atol=1e-4)
    assert abs(loss - lo.item()) < 1e-4
    (lo * S).backward()
    for s in build_layout(cfg)[0]:
        b = P[s.name].grad.numpy().reshape(-1) / S
        a = ga[s.offset:s.offset + s.numel]
        if np.linalg.norm(b) < 1e-9:
            continue
        bound = 2.5e-3 if not s.name.endswith("b_qkv") else 5e-3
        assert np.linalg.norm(a - b) <= bound * np.linalg.norm(b) + 1e-8, (s.name, np.linalg.norm(a - b) / np.linalg.norm(b))


def gemm_args(**kw):
    g = _lib.QstGemmArgs()
    g._keep = [v for v in kw.values() if torch.is_tensor(v)]
    for k, v in kw.items():
        setattr(g, k, v.data_ptr() if torch.is_tensor(v) else v)
    return g


@pytest.mark.parametrize("form", [0, 0x40], ids=["tiled", "eightphase"])
def test_f16_forward_epilogues_saturate_and_backward_ones_overflow(form):
    """sat16 = 1 (what every forward launch of the f16 path sets): results beyond half's range leave as +-65,504; sat16 = 0
    (backward launches): they leave as +-inf, which is what the loss scaler's overflow check looks for."""
    lib = _lib.load()
    M, N, K = 256, 256, 64
    A = torch.zeros(M, K)
    A[:, 0] = 300.0
    B = torch.zeros(N, K)
    B[:, 0] = 300.0                                    # every product 90,000 > 65,504
    B[1::2, 0] = -300.0
    Ad, Bd = A.half().cuda(), B.half().cuda()
    for sat in (1, 0):
        Cd = torch.zeros(M, N, dtype=torch.float16, device="cuda")
        _lib.check(lib.qst_gemm_nt_f16(gemm_args(A=Ad, B=Bd, C=Cd, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, splits=form, sat16=sat), 0,
                                       _lib.current_stream_ptr()))
        c = Cd.float().cpu()
        if sat:
            assert (c[:, 0::2] == 65504.0).all() and (c[:, 1::2] == -65504.0).all()
        else:
            assert torch.isinf(c).all() and (c[:, 0::2] > 0).all() and (c[:, 1::2] < 0).all()
    # GELU epilogue (forward only): h = gelu(u) saturates, gelu'(u) = 1
    bias = torch.full((N,), 1.0e5).cuda()
    C1 = torch.zeros(M, N, dtype=torch.float16, device="cuda")
    C2 = torch.zeros(M, N, dtype=torch.float16, device="cuda")
    Z = torch.zeros(M, K, dtype=torch.float16, device="cuda")
    _lib.check(lib.qst_gemm_nt_f16(gemm_args(A=Z, B=Bd, C=C1, C2=C2, bias=bias, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, splits=form,
                                             sat16=1), 2, _lib.current_stream_ptr()))
    assert (C2.float() == 65504.0).all() and (C1.float() == 1.0).all()


@pytest.mark.parametrize("form", [0, 2, 4], ids=["tiles128", "tiles256", "tall256"])
@pytest.mark.parametrize("M,N,K", [(256, 384, 384), (1000, 1152, 384), (4096, 384, 1536), (300, 1536, 768)])
def test_split_weight_second_pass(M, N, K, form):
    """QstGemmArgs.B2 (QST_PREC_F16W): C = A . (B + B2)^T as a second pass over K in the tiled kernels and in the LayerNorm-fused
    one, with B2 = the low halves of split-f16 weights -- subnormal halves included (the matrix cores keep them) -- against the
    product with the UNSPLIT fp32 weights: the weight rounding is gone (error ~2^-22 per weight), what remains is A's."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).half()
    W = torch.randn(N, K, generator=g) * 0.04
    Wh = W.half()
    Wl = (W - Wh.float()).half()
    assert (Wl.float().abs() < 6.1e-5).float().mean() > 0.9                  # nearly all low halves are subnormal
    bias, resid = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = A.double() @ W.double().t() + bias.double() + resid.double()
    ref_hi = A.double() @ Wh.double().t() + bias.double() + resid.double()
    Ad, Bh, Bl = A.cuda(), Wh.cuda(), Wl.cuda()
    C = torch.empty(M, N, device="cuda")
    _lib.check(lib.qst_gemm_nt_f16(gemm_args(A=Ad, B=Bh, B2=Bl, C=C, bias=bias.cuda(), resid=resid.cuda(), M=M, N=N, K=K, lda=K,
                                             ldb=K, ldc=N, ldr=N, splits=form, sat16=1), 1, _lib.current_stream_ptr()))
    err = (C.double().cpu() - ref).abs().max().item()
    err_hi = (ref_hi - ref).abs().max().item()
    assert err < 0.05 * err_hi + 2e-6 * math.sqrt(K), (err, err_hi)
    if N == 384:
        e = _lib.QstLnEpi()
        gamma, beta = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
        e.gamma, e.beta, e.eps = gamma.data_ptr(), beta.data_ptr(), 1e-12
        Y = torch.empty(M, N, device="cuda")
        _lib.check(lib.qst_gemm_nt_ln_f16(gemm_args(A=Ad, B=Bh, B2=Bl, C=Y, bias=bias.cuda(), resid=resid.cuda(), M=M, N=N, K=K,
                                                    lda=K, ldb=K, ldc=N, ldr=N), e, 0, _lib.current_stream_ptr()))
        yref = torch.nn.functional.layer_norm(ref.float(), (N,), None, None, 1e-12)
        torch.testing.assert_close(Y.cpu(), yref, rtol=1e-4, atol=2e-5)


def test_f16_encoder_survives_activations_beyond_half_range():
    """A feed-forward unit driven to 1e5 (bias b_1) and a LayerNorm gain of 5e3: gelu(u) and the normalised rows exceed 65,504
    in fp32; the f16 operand copies saturate, nothing becomes inf / nan, and the embeddings stay those of the fp32-class path
    run on the same weights with the same clamp in mind (the unit's contribution is a constant vector before LayerNorm)."""
    cfg = PRESETS["tiny-bert"]
    arena = synthetic_params(cfg, seed=14, std=0.05, bias_std=0.02, ln_jitter=0.05)
    segs = {s.name: s for s in build_layout(cfg)[0]}
    b1 = segs["layer.0.b_1"]
    arena[b1.offset + 3] = 1.0e5                                   # u[:, 3] ~ 1e5 -> h[:, 3] = 1e5 (fp32), 65,504 (f16 operand)
    ids, mask, types = synthetic_quadruplets(cfg, 2, 32, seed=14, ragged=True)
    loss, e4, ga, enc = run_f16(cfg, arena, ids, mask, types, 2, 32, scale=1024.0)
    assert np.isfinite(e4).all() and np.isfinite(loss) and np.isfinite(ga).all()
    # the oracle with the operand clamp stated explicitly: h is rounded to f16 WITH saturation before FFN-2
    P = R.arena_to_dict(arena, cfg)
    import torch.nn.functional as F

    def sat16(x):
        return x.clamp(-65504.0, 65504.0).half().double()
    with torch.no_grad():
        idt, mt, tt = [torch.from_numpy(x).view(8, 32) for x in (ids, mask, types)]
        H, A = cfg.hidden_size, cfg.num_heads
        d = H // A
        x = P["word_emb"][idt] + P["type_emb"][tt] + P["pos_emb"][torch.arange(32)][None]
        x = F.layer_norm(x, (H,), P["emb_ln_g"], P["emb_ln_b"], cfg.layer_norm_eps)
        am = (1.0 - mt[:, None, None, :].float()) * torch.finfo(torch.float32).min
        for l in range(cfg.num_layers):
            p = f"layer.{l}."
            lin = lambda a, w, b: (sat16(a) @ sat16(P[p + w]).t()).float() + P[p + b]      # noqa: E731
            qkv = lin(x, "w_qkv", "b_qkv")
            q, k, v = [t.view(8, 32, A, d).transpose(1, 2) for t in qkv.split(H, dim=-1)]
            s = (sat16(q) @ sat16(k).transpose(-1, -2)).float() / math.sqrt(d) + am
            ctx = (sat16(torch.softmax(s, -1)) @ sat16(v)).float().transpose(1, 2).reshape(8, 32, H)
            x = F.layer_norm(lin(ctx, "w_o", "b_o") + x, (H,), P[p + "ln1_g"], P[p + "ln1_b"], cfg.layer_norm_eps)
            h = F.gelu(lin(x, "w_1", "b_1"))
            x = F.layer_norm(lin(h, "w_2", "b_2") + x, (H,), P[p + "ln2_g"], P[p + "ln2_b"], cfg.layer_norm_eps)
        ref = R.st_head(x, mt, cfg.normalize).view(4, 2, -1).numpy()
    np.testing.assert_allclose(e4, ref, rtol=1e-3, atol=2e-4)


def test_f16_gradients_survive_a_512_quadruplet_mean_loss():
    """B = 512, mean-reduced: d(loss)/d(embedding) entries are ~1e-4 and the token-level gradients of the lower layers ~1e-8 --
    below half's smallest normal (6.1e-5) and, unscaled, mostly below its smallest subnormal (6e-8). Under GradScaler's
    initial scale (65536) the f16 backward agrees with the fp32-class backward (bf16x3) to 5e-3 per tensor; without a scale it
    does not (asserted, so that the test would notice a scale that is silently dropped)."""
    cfg = PRESETS["tiny-bert"]
    B, L = 512, 32
    arena = synthetic_params(cfg, seed=14, std=0.05, bias_std=0.02, ln_jitter=0.05)
    ids, mask, types = synthetic_quadruplets(cfg, B, L, seed=14, ragged=True)
    _, _, g_scaled, enc = run_f16(cfg, arena, ids, mask, types, B, L, scale=65536.0)
    _, _, g_plain, _ = run_f16(cfg, arena, ids, mask, types, B, L, scale=None)
    n = 4 * B
    idd, mdd, tdd = [torch.from_numpy(x).view(n, L).cuda() for x in (ids, mask, types)]
    emb, _, saved = enc.forward(idd, mdd, tdd, training=True, precision="bf16x3")
    e4 = emb.view(4, B, -1)
    _, g = quadruplet_loss_raw(e4[0], e4[1], e4[2], e4[3], 0.6, 1.0, 0.5, 0.5, 2.0, False, 2, want_grads=True)
    enc.grads.zero_()
    enc.backward(idd, mdd, tdd, stacked(g), saved, precision="bf16x3")
    ref = enc.grads.cpu().numpy()
    worst_scaled, worst_plain = 0.0, 0.0
    for s in build_layout(cfg)[0]:
        b = ref[s.offset:s.offset + s.numel]
        if np.linalg.norm(b) < 1e-9 or s.name.endswith("b_qkv"):
            continue
        es = np.linalg.norm(g_scaled[s.offset:s.offset + s.numel] - b) / np.linalg.norm(b)
        ep = np.linalg.norm(g_plain[s.offset:s.offset + s.numel] - b) / np.linalg.norm(b)
        assert es <= 5e-3, (s.name, es)
        if s.gemm:
            nz = (np.abs(b) > 1e-3 * np.abs(b).max())
            assert (g_scaled[s.offset:s.offset + s.numel][nz] != 0).mean() > 0.999, s.name
        worst_scaled, worst_plain = max(worst_scaled, es), max(worst_plain, ep)
    assert worst_plain > 4 * worst_scaled, (worst_plain, worst_scaled)


def test_amp_step_follows_gradscaler_rules():
    """qst_clip_adamw_step_amp against torch.optim.AdamW + the rules of torch.cuda.amp.GradScaler as ST's fit() applies them:
    unscale, clip on the unscaled norm, step; an inf anywhere skips the step (parameters, moments, optimiser step count
    untouched; gradients zeroed), halves the scale and -- the scale having changed -- skips scheduler.step(); growth_interval
    good steps in a row double the scale, and that step too skips the scheduler."""
    cfg = PRESETS["tiny-bert"]
    enc = HipEncoder(cfg)
    arena = synthetic_params(cfg, seed=3, std=0.05, bias_std=0.02, ln_jitter=0.05)
    enc.load_arena(arena)
    enc.ensure_train_state()
    enc.ensure_amp_scaler(1024.0)
    segs, total = build_layout(cfg)
    lr, warmup, tot, wd = 1e-3, 2, 10, 0.01
    p = torch.from_numpy(np.asarray(arena)).clone().requires_grad_(True)
    decay = torch.zeros(total, dtype=torch.bool)
    for s in segs:
        decay[s.offset:s.offset + s.numel] = bool(s.decay)
    gen = torch.Generator().manual_seed(0)
    m = torch.zeros(total)
    v = torch.zeros(total)
    opt_t, sched_k, scale, tracker = 0, 0, 1024.0, 0
    interval = 3
    for it in range(8):
        g = torch.randn(total, generator=gen) * 0.01
        overflow = it in (2, 5)
        gs = g * scale
        if overflow:
            gs[12345 % total] = float("inf")
        enc.grads.copy_(gs.cuda())
        before = enc.params.clone()
        enc.adamw_step_amp(lr, warmup, tot, (0.9, 0.999), 1e-8, wd, 1.0, 1.0, growth_interval=interval)
        st = enc.amp_scaler.cpu().numpy()
        cnt = enc._step2_dev.cpu().numpy()
        assert float(enc.grads.abs().max().item()) == 0.0
        cur_lr = warmup_linear_lr(lr, sched_k, warmup, tot)
        if overflow:
            assert torch.equal(enc.params, before)
            scale, tracker = scale * 0.5, 0
            assert st[2] == 1.0
        else:
            opt_t += 1
            norm = g.norm().item()
            gg = g * min(1.0, 1.0 / (norm + 1e-6))
            pp = p.detach().clone()
            pp[decay] *= (1.0 - cur_lr * wd)
            m = m + 0.1 * (gg - m)
            v = 0.999 * v + 0.001 * gg * gg
            denom = v.sqrt() / math.sqrt(1 - 0.999 ** opt_t) + 1e-8
            pp = pp - (cur_lr / (1 - 0.9 ** opt_t)) * (m / denom)
            p = pp
            np.testing.assert_allclose(enc.params.cpu().numpy(), p.numpy(), rtol=2e-5, atol=2e-7)
            np.testing.assert_allclose(enc.grad_norm.item(), norm, rtol=1e-4)
            tracker += 1
            if tracker == interval:
                scale, tracker = scale * 2.0, 0
            assert st[2] == 0.0
        if not (overflow or tracker == 0):
            sched_k += 1                                   # ST: scheduler.step() only when the scale did not change
        assert st[0] == scale and st[1] == tracker, (it, st, scale, tracker)
        assert cnt[0] == opt_t and cnt[1] == sched_k, (it, cnt, opt_t, sched_k)
    assert enc.amp_scaler[3].item() == 2.0
    # a static scale: growth_interval <= 0 never changes it; an overflow still skips
    enc2 = HipEncoder(cfg)
    enc2.load_arena(arena)
    enc2.ensure_train_state()
    enc2.ensure_amp_scaler(256.0)
    enc2.grads.fill_(float("nan"))
    b2 = enc2.params.clone()
    enc2.adamw_step_amp(lr, 0, 0, growth_interval=0)
    assert torch.equal(enc2.params, b2) and enc2.amp_scaler[0].item() == 256.0 and enc2.grads.abs().max().item() == 0.0


@pytest.mark.parametrize("name,drop,graph,prec", [("tiny-bert", None, False, "f16"), ("tiny-mpnet", None, False, "f16"),
                                                  ("tiny-bert", 0.1, False, "f16"), ("tiny-bert", None, True, "f16"),
                                                  ("tiny-mpnet", 0.1, False, "f16w"), ("tiny-bert", None, True, "f16w")])
def test_f16_training_tracks_the_fp32_reference(name, drop, graph, prec):
    """QuadrupletTrainer(precision="f16") -- f16 operands, dynamic loss scale on the device -- against the fp32 oracle (torch
    autograd + torch.optim.AdamW, no operand rounding): six steps, the loss within 5e-4 of the reference's at every step (the
    bf16 path is held to 3e-3 against an oracle that rounds like it; bf16x3 to 1e-4) and the parameters within 6% of the
    distance training moved them. graph: the same step captured into a HIP graph (schedule, step counters and scaler all live
    on the device)."""
    cfg = PRESETS[name]
    B, L, steps, lr, warmup, total = 6, 32, 6, 2e-3, 2, 20
    arena = synthetic_params(cfg, seed=14, std=0.05, bias_std=0.02, ln_jitter=0.05)
    P = R.arena_to_dict(arena, cfg, requires_grad=True)
    segs, _ = build_layout(cfg)
    groups = [{"params": [P[s.name] for s in segs if s.decay], "weight_decay": 0.01},
              {"params": [P[s.name] for s in segs if not s.decay], "weight_decay": 0.0}]
    opt = torch.optim.AdamW(groups, lr=lr, betas=(0.9, 0.999), eps=1e-8)
    tr = QuadrupletTrainer(cfg, arena=arena, device="cuda:0", lr=lr, weight_decay=0.01, max_grad_norm=1.0, warmup_steps=warmup,
                           total_steps=total, precision=prec, dropout=drop, dropout_seed=31, use_graph=graph, **LOSS_KW)
    ref_losses, hip_losses = [], []
    for step in range(steps):
        ids, mask, types = synthetic_quadruplets(cfg, B, L, seed=14, ragged=True, step=0)
        t = [torch.from_numpy(x) for x in (ids, mask, types)]
        for g in opt.param_groups:
            g["lr"] = warmup_linear_lr(lr, step, warmup, total)
        opt.zero_grad()
        masks = None
        if drop:
            from oracle.dropout_ref import Masks
            masks = Masks(31, step + 1, drop, drop)
        loss, _ = R.quadruplet_step(P, cfg, *t, LOSS_KW, bf16_operands=False, dropout=masks)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([p for g in opt.param_groups for p in g["params"]], 1.0)
        opt.step()
        ref_losses.append(loss.item())
        hip_losses.append(tr.step(*[x.cuda() for x in t]).item())
    ref_losses, hip_losses = np.array(ref_losses), np.array(hip_losses)
    assert ref_losses[-1] < ref_losses[0] - 0.03, "reference did not train"
    np.testing.assert_allclose(hip_losses, ref_losses, rtol=0, atol=5e-4)
    assert tr.enc.amp_scaler[0].item() == 65536.0 and tr.enc.amp_scaler[3].item() == 0.0       # no step overflowed
    got = tr.enc.params.cpu().numpy()
    for s in segs:
        a = got[s.offset:s.offset + s.numel]
        b = P[s.name].detach().numpy().reshape(-1)
        w0 = arena[s.offset:s.offset + s.numel]
        if s.name.endswith("b_qkv"):
            H = cfg.hidden_size
            keep = np.r_[0:H, 2 * H:3 * H]          # (the key third: a zero gradient, Adam steps of random sign in both)
            a, b, w0 = a[keep], b[keep], w0[keep]
        moved = np.abs(b - w0).mean()
        err = np.abs(a - b).mean()
        assert err <= 0.06 * moved + 1e-7, f"{s.name}: mean |diff| {err:.3e} vs mean |update| {moved:.3e}"


def test_f16_handles_refuse_each_others_arenas():
    """An activation arena filled by a bf16 training forward is refused by the f16 backward and the reverse (QST_ERR_NO_FORWARD):
    the 16-bit tensors in it are of the other type."""
    cfg = PRESETS["tiny-bert"]
    enc = HipEncoder(cfg)
    enc.load_arena(synthetic_params(cfg, seed=14, std=0.05))
    ids, mask, types = synthetic_quadruplets(cfg, 2, 32, seed=14, ragged=True)
    idd, mdd, tdd = [torch.from_numpy(x).view(8, 32).cuda() for x in (ids, mask, types)]
    ge = torch.zeros(8, cfg.hidden_size, device="cuda")
    enc.ensure_train_state()
    for fwd, bwd in (("bf16", "f16"), ("f16", "bf16")):
        _, _, saved = enc.forward(idd, mdd, tdd, training=True, precision=fwd)
        with pytest.raises(_lib.QstError, match="no matching training forward"):
            enc.backward(idd, mdd, tdd, ge, saved, precision=bwd)
        enc.backward(idd, mdd, tdd, ge, saved, precision=fwd)
    # ... and so is a backward with another shape than the forward that filled the arena (the record holds nseq and L)
    _, _, saved = enc.forward(idd, mdd, tdd, training=True, precision="bf16")
    with pytest.raises(_lib.QstError, match="no matching training forward"):
        enc.backward(idd.view(4, 64), mdd.view(4, 64), tdd.view(4, 64), ge[:4], saved, precision="bf16")


def check_against_fp32_autograd(name, B, L, ragged, wkw, drop, prec, layers=None, grad_bound=5e-3):
    """forward(training=True) + backward at precision `prec` ("f16" / "f16w"), under GradScaler's initial loss scale, against fp32
    torch autograd with the same dropout masks (oracle/dropout_ref.py): embeddings at the north-star tolerance for "f16w" and at
    twice its atol for plain "f16" (whose weight rounding, DESIGN.md finding 34, puts single elements of trained-like models at
    the edge), both scaled by 1 / (1 - p_hidden) in train() mode; every gradient tensor within `grad_bound` relative L2 on the
    scale of its class. Also the body of tools/fuzz_shapes.py's f16 / f16w modes."""
    from dataclasses import replace
    from tests.test_gpu_encoder import cls_of
    if name == "mpnet-2l":
        cfg = replace(PRESETS["all-mpnet-base-v2"], num_layers=2, vocab_size=4096)
    else:
        cfg = PRESETS[name] if layers is None else replace(PRESETS[name], num_layers=layers, vocab_size=2048)
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
    emb, _, saved = enc.forward(idd, mdd, tdd, training=True, precision=prec)
    e4 = emb.view(4, B, -1)
    S = 65536.0
    loss, g = quadruplet_loss_raw(e4[0], e4[1], e4[2], e4[3], 0.6, 1.0, 0.5, 0.5, 2.0, False, 2,
                                  grad_out=torch.tensor([S], device="cuda"), want_grads=True)
    enc.grads.zero_()
    enc.backward(idd, mdd, tdd, stacked(g), saved, precision=prec)
    torch.cuda.synchronize()
    k = 1.0 / (1.0 - (drop[0] if drop is not None else 0.0))
    sc = max(1.0, float(emb32.detach().norm(dim=-1).mean()))                   # (bare bert-base emits un-normalised embeddings)
    d = (emb.cpu().view(4, B, -1) - emb32.detach()).abs()
    atol = (1e-4 if prec == "f16w" else 2e-4) * k * sc
    bad = d > atol + 1e-3 * emb32.detach().abs()
    print(f"[{prec} fwd] {name} B={B} L={L} drop={drop}: max|d emb| {float(d.max()):.2e} (atol {atol:.1e}), |d loss| {abs(loss.item() - loss32.item()):.1e}")
    assert not bool(bad.any()), (float(d.max()), atol, int(bad.sum()))
    assert abs(loss.item() - loss32.item()) < 1e-3 * k * sc
    # a hinge within the forward tolerance of its kink is on in one implementation and off in the other (tests/test_gpu_fp8mx.py)
    eo = emb32.detach()
    dist = lambda x, y: (x - y + 1e-6).norm(dim=-1)                            # noqa: E731
    args = torch.stack([1.0 + dist(eo[0], eo[1]) - dist(eo[0], eo[3]), 0.5 + dist(eo[0], eo[2]) - dist(eo[0], eo[3]),
                        0.5 + dist(eo[0], eo[1]) - dist(eo[0], eo[2])])
    if float(args.abs().min()) < 4 * atol:
        print(f"[{prec}] a hinge within {float(args.abs().min()):.1e} of its kink -- gradients not compared")
        return
    segs, _ = build_layout(cfg)
    ga = enc.grads.cpu() / S
    assert torch.isfinite(ga).all()
    gnorm = float(torch.sqrt(sum((P[s_.name].grad.double() ** 2).sum() for s_ in segs)))
    cls_top = {}
    for s_ in segs:
        c = cls_of(s_.name.split(".")[-1])
        cls_top[c] = max(cls_top.get(c, 0.0), P[s_.name].grad.norm().item())
    worst = (0.0, "")
    for s_ in segs:
        ref = P[s_.name].grad
        got = ga[s_.offset:s_.offset + s_.numel].view(*s_.shape)
        denom = ref.norm().item()
        if denom <= 1e-5 * gnorm:
            assert got.norm().item() <= 1e-4 * gnorm, s_.name
            continue
        err = ((got - ref).norm() / max(denom, 0.05 * cls_top[cls_of(s_.name.split(".")[-1])])).item()
        worst = max(worst, (err, s_.name))
        lim = grad_bound * k * (2.0 if s_.name.endswith("b_qkv") else 1.0)
        assert err < lim, f"{name} grad {s_.name}: relative L2 error {err:.3e} against {lim:.1e} (ref norm {denom:.3e})"
    print(f"[{prec} grad-err] {name} B={B} L={L}: worst {worst[1]} {worst[0]:.2e}")


@pytest.mark.parametrize("prec", ["f16", "f16w"])
@pytest.mark.parametrize("name,B,L,ragged,wkw,drop", [
    ("tiny-mpnet", 2, 64, True, dict(std=0.08, bias_std=0.05, ln_jitter=0.1), None),
    ("minilm-2l", 2, 128, True, dict(std=0.03, bias_std=0.02, ln_jitter=0.05), (0.1, 0.1, 6)),
    ("mpnet-2l", 1, 288, True, dict(std=0.03, bias_std=0.02, ln_jitter=0.05), None),          # d = 64, L > 256: one-workgroup backward
    ("mpnet-2l", 1, 512, True, dict(std=0.03, bias_std=0.02, ln_jitter=0.05), (0.2, 0.1, 7)),  # L = 512 + position bias: the dQ / dK,dV pair
    ("minilm-2l", 33, 128, True, dict(std=0.03, bias_std=0.02, ln_jitter=0.05), (0.1, 0.1, 9))])   # M = 16,896: the LayerNorm-fused kernels
def test_f16_forward_and_backward_against_fp32_autograd(name, B, L, ragged, wkw, drop, prec):
    check_against_fp32_autograd(name, B, L, ragged, wkw, drop, prec)


def test_f16_training_recovers_from_an_overflowing_loss_scale():
    """GradScaler's init_scale far too large (2^30: d(loss)/d(embedding) x 2^30 is beyond half's range, so the first f16 gradient
    tensors of the backward are inf and everything behind them inf or nan): every such step must be SKIPPED -- parameters and
    moments untouched, gradients zeroed, scale halved -- until the scale fits, and training must then proceed as if nothing had
    happened (a later step whose gradients outgrow the scale is skipped the same way). The whole decision runs on the device
    (qst_clip_adamw_step_amp); no inf / nan may reach the parameters."""
    cfg = PRESETS["tiny-bert"]
    arena = synthetic_params(cfg, seed=14, std=0.05, bias_std=0.02, ln_jitter=0.05)
    tr = QuadrupletTrainer(cfg, arena=arena, device="cuda:0", lr=2e-3, weight_decay=0.01, max_grad_norm=1.0, warmup_steps=0,
                           total_steps=0, precision="f16", amp_init_scale=2.0 ** 30, **LOSS_KW)
    ids, mask, types = synthetic_quadruplets(cfg, 6, 32, seed=14, ragged=True, step=0)
    batch = [torch.from_numpy(x).cuda() for x in (ids, mask, types)]
    p0 = tr.enc.params.clone()
    losses, scales, skipped = [], [], []
    nsteps = 24
    for _ in range(nsteps):
        losses.append(tr.step(*batch).item())
        st = tr.enc.amp_scaler.cpu().tolist()
        scales.append(st[0])
        skipped.append(int(st[3]))
        assert torch.isfinite(tr.enc.params).all() and torch.isfinite(tr.enc.exp_avg).all() and torch.isfinite(tr.enc.exp_avg_sq).all()
        assert float(tr.enc.grads.abs().max()) == 0.0                          # zero_grad, skipped or not
        if skipped[-1] == len(losses):                       # every step so far was skipped: nothing may have moved
            assert torch.equal(tr.enc.params, p0)
    nskip = skipped[-1]
    first = next(i for i in range(nsteps) if skipped[i] <= i)              # the first step that was NOT skipped
    assert 4 <= first <= 12, (first, scales)
    assert scales[-1] == 2.0 ** 30 / 2 ** nskip                # one halving per skipped step, no growth within 24 steps
    cnt = tr.enc._step2_dev.cpu().tolist()
    assert cnt[0] == nsteps - nskip                              # optimiser steps taken = steps that were not skipped
    assert all(abs(l - losses[0]) < 1e-6 for l in losses[:first + 1])      # (same batch, unchanged parameters: the same loss)
    assert losses[-1] < losses[first] - 0.05, losses             # ... and it trains from there
