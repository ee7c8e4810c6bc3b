"""Dropout on the training path (include/qst_kernels.h: QstDrop; reference: HF modules in train() mode under
SentenceTransformer.fit, /root/reference/training/main.py:128). The kernels recompute counter-based masks instead of storing
them; oracle/dropout_ref.py regenerates the same masks in numpy, so every check below is the usual comparison against the
torch oracle, run with identical masks: the mask words bit for bit, each kernel that applies one, attention forward and
backward on every code path, whole encoders with gradients, and the life cycle (several forwards alive before their
backwards, checkpoint resume, inference never drops)."""
import math
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from oracle import dropout_ref as D  # noqa: E402
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402
from quadruplet_sentence_transformer_amd.config import PRESETS  # noqa: E402
from quadruplet_sentence_transformer_amd.encoder import HipEncoder  # noqa: E402
from quadruplet_sentence_transformer_amd.synthetic import synthetic_params, synthetic_quadruplets  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


def stream():
    return _lib.current_stream_ptr()


def make_state(lib, seed, step):
    st = torch.zeros(4, dtype=torch.int32, device="cuda")
    _lib.check(lib.qst_dropout_init(st.data_ptr(), seed, stream()))
    for _ in range(step):
        _lib.check(lib.qst_dropout_advance(st.data_ptr(), stream()))
    return st


def drop(st, site, p):
    d = _lib.QstDrop()
    d.state, d.site, d.thr16 = st.data_ptr(), site, D.thr16_of(p)
    return d


def mult(seed, step, site, shape, p):
    return torch.from_numpy(D.multipliers(seed, step, site, int(np.prod(shape)), p).reshape(shape))


@pytest.mark.parametrize("seed,step,site,n,p", [(0, 0, 0, 1000, 0.1), (1234567890123456789, 3, D.SITE_EMBED, 99999, 0.1),
                                                (7, 1000, D.site_probs(11), 1 << 20, 0.37), (2 ** 40 + 5, 2, 5, 7, 0.5)])
def test_mask_words_equal_the_oracle(lib, seed, step, site, n, p):
    st = make_state(lib, seed, step)
    out = torch.empty(n, device="cuda")
    _lib.check(lib.qst_dropout_multipliers(drop(st, site, p), 0, n, out.data_ptr(), stream()))
    assert np.array_equal(out.cpu().numpy(), D.multipliers(seed, step, site, n, p))
    _lib.check(lib.qst_dropout_multipliers(drop(st, site, p), 1, n, out.data_ptr(), stream()))      # the attention-probability form
    assert np.array_equal(out.cpu().numpy(), D.multipliers8(seed, step, site, n, p))
    assert st.cpu().tolist()[2] == step


def bfr(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("M,H", [(37, 64), (300, 384), (129, 768)])
def test_row_kernels_apply_the_mask(lib, M, H):
    g = torch.Generator().manual_seed(M + H)
    seed, step, p = 99, 2, 0.2
    st = make_state(lib, seed, step)
    mk = mult(seed, step, D.SITE_EMBED, (M, H), p).cuda()
    # embeddings + LayerNorm + dropout
    V, P = 50, 40
    ids = torch.randint(0, V, (M,), generator=g).cuda(); pos = torch.randint(0, P, (M,), generator=g).int().cuda()
    word = torch.randn(V, H, generator=g).cuda(); pemb = torch.randn(P, H, generator=g).cuda()
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).cuda(); beta = (0.1 * torch.randn(H, generator=g)).cuda()
    outs = []
    for d in (None, drop(st, D.SITE_EMBED, p)):
        y = torch.empty(M, H, device="cuda"); yb = torch.empty(M, H, dtype=torch.bfloat16, device="cuda")
        xh = torch.empty(M, H, dtype=torch.bfloat16, device="cuda"); rs = torch.empty(M, device="cuda")
        _lib.check(lib.qst_embed_ln_fwd_drop(ids.data_ptr(), None, pos.data_ptr(), word.data_ptr(), pemb.data_ptr(), None,
                                             gamma.data_ptr(), beta.data_ptr(), 1e-12, M, H, y.data_ptr(), yb.data_ptr(),
                                             xh.data_ptr(), rs.data_ptr(), d, stream()))
        outs.append((y, yb, xh, rs))
    (y0, yb0, xh0, rs0), (y1, yb1, xh1, rs1) = outs
    assert torch.equal(y1, y0 * mk) and torch.equal(yb1, (y0 * mk).to(torch.bfloat16))
    assert torch.equal(xh0, xh1) and torch.equal(rs0, rs1)                     # the saved normalised row is the undropped one
    # LayerNorm backward: mask on the incoming gradient (a dropout after the LayerNorm) / on the bf16 result only
    dy = torch.randn(M, H, generator=g).cuda()
    res = []
    for dy_in, din, dout in ((dy, None, None), (dy * mk, None, None), (dy, drop(st, D.SITE_EMBED, p), None),
                             (dy, None, drop(st, D.SITE_EMBED, p))):
        ds = torch.empty(M, H, device="cuda"); dsb = torch.empty(M, H, dtype=torch.bfloat16, device="cuda")
        dg = torch.zeros(H, device="cuda"); db = torch.zeros(H, device="cuda")
        _lib.check(lib.qst_ln_bwd_drop(dy_in.data_ptr(), xh0.data_ptr(), rs0.data_ptr(), gamma.data_ptr(), M, H, ds.data_ptr(),
                                       dsb.data_ptr(), dg.data_ptr(), db.data_ptr(), None, din, dout, stream()))
        torch.cuda.synchronize()
        res.append((ds, dsb, dg, db))
    plain, premasked, masked_in, masked_out = res
    for a, b in zip(premasked[:2], masked_in[:2]):
        assert torch.equal(a, b)
    for a, b in zip(premasked[2:], masked_in[2:]):                                 # float atomics
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5 * max(1.0, a.abs().max().item()))
    assert torch.equal(masked_out[0], plain[0]) and torch.equal(masked_out[1], (plain[0] * mk).to(torch.bfloat16))


@pytest.mark.parametrize("M,K,N", [(300, 384, 384), (1000, 1536, 384), (700, 768, 768), (1300, 3072, 768), (33100, 128, 768)])
def test_projection_epilogues_apply_the_mask(lib, M, K, N):
    """C = (A.B^T + bias) * mask + resid: the plain fp32 epilogue and the LayerNorm-fused one (forward; N = 768: the
    several-tiles-per-row form); the fused LayerNorm backward with the mask on its bf16 result (where 2) or on the incoming
    gradient (where 3)."""
    g = torch.Generator().manual_seed(M + K)
    seed, step, p, site = 5, 1, 0.1, D.site_ffn_out(3)
    st = make_state(lib, seed, step)
    mk = mult(seed, step, site, (M, N), p)
    A = bfr(torch.randn(M, K, generator=g)); B = bfr(torch.randn(N, K, generator=g) * 0.05)
    bias = torch.randn(N, generator=g); resid = torch.randn(M, N, generator=g)
    gamma = 1 + 0.1 * torch.randn(N, generator=g); beta = 0.1 * torch.randn(N, generator=g)
    Ad, Bd = A.to(torch.bfloat16).cuda(), B.to(torch.bfloat16).cuda()
    biasd, residd, gd, bd = bias.cuda(), resid.cuda(), gamma.cuda(), beta.cuda()

    def args(**kw):
        ga = _lib.QstGemmArgs()
        ga.A, ga.B, ga.M, ga.N, ga.K, ga.lda, ga.ldb, ga.ldc, ga.ldr = Ad.data_ptr(), Bd.data_ptr(), M, N, K, K, K, N, N
        ga.resid = residd.data_ptr()
        for k, v in kw.items():
            setattr(ga, k, v)
        return ga
    v = (A @ B.t() + bias) * mk + resid
    C = torch.empty(M, N, device="cuda")
    _lib.check(lib.qst_gemm_nt(args(C=C.data_ptr(), bias=biasd.data_ptr(), drop=drop(st, site, p), drop_where=1), 1, stream()))
    torch.testing.assert_close(C.cpu(), v, rtol=1e-4, atol=1e-3)
    # fused forward LayerNorm
    ln = _lib.QstLnEpi()
    xh = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"); rs = torch.empty(M, device="cuda")
    C2 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ln.gamma, ln.beta, ln.eps, ln.xhat, ln.rstd = gd.data_ptr(), bd.data_ptr(), 1e-12, xh.data_ptr(), rs.data_ptr()
    _lib.check(lib.qst_gemm_nt_ln(args(C=C.data_ptr(), C2=C2.data_ptr(), bias=biasd.data_ptr(), drop=drop(st, site, p), drop_where=1),
                                  ln, 0, stream()))
    ref = torch.nn.functional.layer_norm(v, (N,), gamma, beta, 1e-12)
    torch.testing.assert_close(C.cpu(), ref, rtol=1e-3, atol=2e-3)
    # fused LayerNorm backward
    xhat = bfr(torch.randn(M, N, generator=g)); rstd = torch.rand(M, generator=g) + 0.5
    xhd, rsd = xhat.to(torch.bfloat16).cuda(), rstd.cuda()
    ln2 = _lib.QstLnEpi()
    br = lib.qst_gemm_nt_ln_block_rows_m(N, M)
    part = torch.zeros((M + br - 1) // br, 2, N, device="cuda")
    ln2.gamma, ln2.xhat, ln2.rstd, ln2.partials = gd.data_ptr(), xhd.data_ptr(), rsd.data_ptr(), part.data_ptr()

    def ln_bwd_ref(dy):
        dx = dy * gamma
        return rstd[:, None] * (dx - dx.mean(1, keepdim=True) - xhat * (dx * xhat).mean(1, keepdim=True))
    dy = A @ B.t() + resid
    for where, want_c, want_c2, want_dgamma in ((2, ln_bwd_ref(dy), ln_bwd_ref(dy) * mk, (dy * xhat).sum(0)),
                                                (3, ln_bwd_ref(dy * mk), ln_bwd_ref(dy * mk), (dy * mk * xhat).sum(0))):
        _lib.check(lib.qst_gemm_nt_ln(args(C=C.data_ptr(), C2=C2.data_ptr(), drop=drop(st, site, p), drop_where=where), ln2, 1, stream()))
        torch.testing.assert_close(C.cpu(), want_c, rtol=1e-3, atol=2e-3)
        torch.testing.assert_close(C2.float().cpu(), want_c2, rtol=1e-2, atol=1e-2)
        assert ((C2.float().cpu() == 0) == (want_c2 == 0)).float().mean() > 0.999      # zeros exactly where the mask is (where 2)
        torch.testing.assert_close(part[:, 0].sum(0).cpu(), want_dgamma, rtol=1e-3, atol=1e-3 * math.sqrt(M) * 4)
    # a mask description the kernel cannot honour is refused
    assert lib.qst_gemm_nt(args(C=C.data_ptr(), drop=drop(st, site, p), drop_where=2), 1, stream()) != 0
    assert lib.qst_gemm_nt_ln(args(C=C.data_ptr(), drop=drop(st, site, p), drop_where=1), ln2, 1, stream()) != 0


def attn_ref(qkv, mask, rel, n, L, A, d, pm):
    H = A * d
    q, k, v = [t.view(n, L, A, d).transpose(1, 2) for t in qkv.view(n, L, 3 * H).split(H, dim=-1)]
    s = q @ k.transpose(-1, -2) / math.sqrt(d)
    if rel is not None:
        s = s + rel[None]
    s = s + (1.0 - mask[:, None, None, :].float()) * torch.finfo(torch.float32).min
    return ((torch.softmax(s, -1) * pm) @ v).transpose(1, 2).reshape(n * L, H)


@pytest.mark.parametrize("n,L,A,d,use_rel", [(2, 32, 2, 32, False), (3, 128, 12, 32, False), (2, 160, 2, 64, True),
                                              (2, 256, 3, 64, False), (2, 64, 2, 32, True), (2, 160, 2, 32, True),
                                              (3, 128, 4, 32, False), (2, 96, 2, 64, True)])
def test_attention_with_dropped_probabilities(lib, n, L, A, d, use_rel):
    """Forward and backward on every attention code path (single-workgroup backward: L <= 128, d = 32; dQ + dK/dV kernels
    otherwise; with and without the relative-position bias) against torch autograd with the same mask
    -- and the two backward paths against each other where both apply."""
    H = A * d
    g = torch.Generator().manual_seed(n * L + A + d)
    seed, step, p, site = 31337, 4, 0.1, D.site_probs(1)
    st = make_state(lib, seed, step)
    pm = torch.from_numpy(D.multipliers8(seed, step, site, n * A * L * L, p).reshape(n, A, L, L))
    qkv = bfr(torch.randn(n * L, 3 * H, generator=g))
    lens = torch.randint(max(1, L // 8), L + 1, (n,), generator=g); lens[0] = L
    mask = (torch.arange(L)[None, :] < lens[:, None]).long()
    relpos = (0.5 * torch.randn(A, 2 * L, generator=g)) if use_rel else None
    ridx = (torch.arange(L)[None, :] - torch.arange(L)[:, None]) + L
    dctx = bfr(torch.randn(n * L, H, generator=g))
    qr = qkv.clone().requires_grad_(True)
    relr = relpos.clone().requires_grad_(True) if use_rel else None
    ref = attn_ref(qr, mask, relr[:, ridx] if use_rel else None, n, L, A, d, pm)
    (ref * dctx).sum().backward()

    qd = qkv.to(torch.bfloat16).cuda(); md = mask.cuda(); reld = relpos.cuda() if use_rel else None
    ctx = torch.empty(n * L, H, dtype=torch.bfloat16, device="cuda"); lse = torch.empty(n, A, L, device="cuda")
    q = _lib.QstAttnDesc()
    q.qkv, q.mask, q.rel_pos, q.nseq, q.L, q.A, q.d = qd.data_ptr(), md.data_ptr(), _lib.ptr(reld), n, L, A, d
    q.ctx, q.lse, q.drop = ctx.data_ptr(), lse.data_ptr(), drop(st, site, p)
    _lib.check(lib.qst_attention_fwd_ex(q, stream()))
    torch.testing.assert_close(ctx.float().cpu(), ref.detach(), rtol=2e-2, atol=2e-2)
    # lse is the log-sum-exp of the UNDROPPED scores: same as a forward without dropout
    q0 = _lib.QstAttnDesc.from_buffer_copy(q)
    q0.drop = _lib.QstDrop()
    ctx0 = torch.empty_like(ctx); lse0 = torch.empty_like(lse)
    q0.ctx, q0.lse = ctx0.data_ptr(), lse0.data_ptr()
    _lib.check(lib.qst_attention_fwd_ex(q0, stream()))
    assert torch.equal(lse, lse0) and not torch.equal(ctx, ctx0)

    dq = torch.empty(n * L * 3 * H, dtype=torch.bfloat16, device="cuda")
    drel = torch.zeros(A, 2 * L, device="cuda") if use_rel else None
    dcd = dctx.to(torch.bfloat16).cuda(); delta = torch.empty(n, A, L, device="cuda")
    q.dctx, q.dqkv, q.drel, q.delta_scratch = dcd.data_ptr(), dq.data_ptr(), _lib.ptr(drel), delta.data_ptr()
    _lib.check(lib.qst_attention_bwd_ex(q, stream()))
    gref = qr.grad
    got = dq.view(n * L, 3 * H).float().cpu()
    assert (got - gref).abs().max().item() <= 3e-2 * max(1.0, gref.abs().max().item())
    assert ((got - gref).norm() / gref.norm()).item() < 1e-2
    if use_rel:
        assert ((drel.cpu() - relr.grad).norm() / relr.grad.norm()).item() < 1e-2
    if L <= 128 and d == 32:
        dq2 = torch.empty_like(dq)
        drel2 = torch.zeros(A, 2 * L, device="cuda") if use_rel else None
        q.dqkv, q.drel = dq2.data_ptr(), _lib.ptr(drel2)
        q.force_split = 1
        _lib.check(lib.qst_attention_bwd_ex(q, stream()))
        torch.cuda.synchronize()
        torch.testing.assert_close(dq.float(), dq2.float(), rtol=2e-2, atol=2e-2 * max(1.0, gref.abs().max().item()))


# ------------------------------------------------------------------ whole encoders
from test_gpu_encoder import run_case  # noqa: E402


@pytest.mark.parametrize("name,B,L", [("tiny-bert", 3, 64), ("tiny-mpnet", 2, 64)])
def test_tiny_encoders_train_with_dropout(name, B, L):
    run_case(name, B, L, True, dict(std=0.08, bias_std=0.05, ln_jitter=0.1), emb_atol_vs_bf16_oracle=1.5e-3,
             dropout=(0.1, 0.1, 2024))


def test_minilm_fused_path_trains_with_dropout():
    """M = 16384 rows at MiniLM layer dims: the LayerNorm-fused GEMM epilogues (forward, and both backward mask positions:
    on the bf16 result for the layers, on the incoming gradient for the embedding LayerNorm) and the single-workgroup
    attention backward, all with masks; hidden and attention rates differ so that a swapped threshold would show."""
    from dataclasses import replace
    try:
        run_case("minilm-2l", 32, 128, True, dict(std=0.03, bias_std=0.02, ln_jitter=0.05), emb_atol_vs_bf16_oracle=1.5e-3,
                 dropout=(0.1, 0.15, 7))
    finally:
        pass


def test_fused_path_with_a_ragged_last_tile_and_dropout():
    """M = 4 * 129 * 32 = 16512 token rows: above the fusion threshold and not a multiple of the 128-row tile of the fused
    GEMM+LayerNorm kernels (the last workgroup has 0 rows of its second 64-row half), short sequences (L = 32: four
    sequences per tile), masks on."""
    from dataclasses import replace
    PRESETS["minilm-1l"] = replace(PRESETS["all-MiniLM-L6-v2"], num_layers=1, vocab_size=2048)
    try:
        run_case("minilm-1l", 129, 32, True, dict(std=0.03, bias_std=0.02, ln_jitter=0.05), emb_atol_vs_bf16_oracle=1.5e-3,
                 dropout=(0.1, 0.1, 5))
    finally:
        del PRESETS["minilm-1l"]


def test_long_sequences_and_wide_heads_train_with_dropout():
    """d = 64 heads with the relative-position bias and L = 256 (dQ and dK/dV kernels), unfused LayerNorms (H = 768)."""
    from dataclasses import replace
    PRESETS["mpnet-2l"] = replace(PRESETS["all-mpnet-base-v2"], num_layers=2, vocab_size=4096)
    try:
        run_case("mpnet-2l", 1, 256, True, dict(std=0.02), emb_atol_vs_bf16_oracle=1e-3, dropout=(0.1, 0.1, 11))
    finally:
        del PRESETS["mpnet-2l"]


def test_each_forward_keeps_its_own_masks_until_its_backward():
    """fit() encodes the four columns one after the other and only then runs the backwards: every training forward stores
    the counter value it used next to its activations, so the masks of forward k are rebuilt in backward k whatever ran
    in between. Inference forwards neither drop nor advance the counter."""
    from oracle import torch_ref as R
    cfg = PRESETS["tiny-bert"]
    arena = synthetic_params(cfg, seed=5, std=0.08, bias_std=0.05, ln_jitter=0.1)
    seed, ph, pa = 77, 0.1, 0.1
    enc = HipEncoder(cfg)
    enc.load_arena(arena)
    enc.ensure_train_state()
    enc.set_dropout(ph, pa, seed)
    batches = []
    for k in range(3):
        ids, mask, types = [torch.from_numpy(x).view(-1, 32) for x in synthetic_quadruplets(cfg, 2, 32, seed=20 + k, ragged=True)]
        batches.append((ids, mask, types))
    e_inf = enc.forward(*[t.cuda() for t in batches[0]], training=False)[0].clone()
    live = []
    for ids, mask, types in batches:
        emb, _, saved = enc.forward(ids.cuda(), mask.cuda(), types.cuda(), training=True,
                                    saved=torch.empty(enc.lib.qst_encoder_saved_bytes(enc.handle, ids.shape[0], 32, 1),
                                                      dtype=torch.uint8, device="cuda"))
        live.append((emb.clone(), saved))
    assert enc.dropout_step == 3 and enc.drop_state.cpu().tolist()[2] == 3
    assert torch.equal(e_inf, enc.forward(*[t.cuda() for t in batches[0]], training=False)[0])
    assert not torch.equal(e_inf, live[0][0])
    enc.grads.zero_()
    gens = [torch.randn(live[k][0].shape, generator=torch.Generator().manual_seed(k)) for k in range(3)]
    for k in (2, 0, 1):                                   # backwards in another order than the forwards
        ids, mask, types = batches[k]
        enc.backward(ids.cuda(), mask.cuda(), types.cuda(), gens[k].cuda(), live[k][1])
    torch.cuda.synchronize()
    P = R.arena_to_dict(arena, cfg, requires_grad=True)
    for k, (ids, mask, types) in enumerate(batches):
        emb = R.sentence_embeddings(P, cfg, ids, mask, types, bf16_operands=True, dropout=D.Masks(seed, k + 1, ph, pa))
        torch.testing.assert_close(live[k][0].cpu(), emb.detach(), rtol=1e-3, atol=1.5e-3)
        (emb * gens[k]).sum().backward()
    from quadruplet_sentence_transformer_amd.config import build_layout
    segs, _ = build_layout(cfg)
    ga = enc.grads.cpu()
    for s in segs:
        ref = P[s.name].grad
        if ref.norm().item() < 1e-12:
            continue
        got = ga[s.offset:s.offset + s.numel].view(*s.shape)
        leaf = s.name.split(".")[-1]
        lim = 4e-2 if leaf == "b_qkv" else (3e-2 if (leaf.startswith("b_") or "ln" in leaf) else 1.5e-2)
        assert ((got - ref).norm() / ref.norm()).item() < lim, s.name


def test_backward_uses_the_masks_of_its_forward_whatever_the_handle_is_set_to_since():
    """fit() and bench.py switch dropout on a live encoder. A backward regenerates the masks of the forward that filled its
    activation arena with THAT forward's rates (it took the handle's current setting -- dropout switched off
    between a forward and its backward silently skipped the masks, switched on it read an unwritten snapshot)."""
    cfg = PRESETS["tiny-bert"]
    arena = synthetic_params(cfg, seed=5, std=0.08, bias_std=0.05, ln_jitter=0.1)
    ids, mask, types = [torch.from_numpy(x).view(-1, 32).cuda() for x in synthetic_quadruplets(cfg, 2, 32, seed=21, ragged=True)]
    g = torch.randn(ids.shape[0], cfg.hidden_size, generator=torch.Generator().manual_seed(1)).cuda()

    def run(p_fwd, p_bwd):
        enc = HipEncoder(cfg)
        enc.load_arena(arena)
        enc.ensure_train_state()
        enc.set_dropout(p_fwd, p_fwd, 77)
        nbytes = enc.lib.qst_encoder_saved_bytes(enc.handle, ids.shape[0], 32, 1)
        saved = torch.full((nbytes,), 0xAB, dtype=torch.uint8, device="cuda")      # poisoned: an unwritten snapshot would show
        emb, _, saved = enc.forward(ids, mask, types, training=True, saved=saved)
        enc.set_dropout(p_bwd, p_bwd, 77)                                            # ... changed before the backward
        enc.grads.zero_()
        enc.backward(ids, mask, types, g, saved)
        torch.cuda.synchronize()
        return emb.clone(), enc.grads.clone()

    e_on, g_on = run(0.1, 0.1)
    e_on2, g_on_then_off = run(0.1, 0.0)
    e_off, g_off = run(0.0, 0.0)
    e_off2, g_off_then_on = run(0.0, 0.1)
    assert torch.equal(e_on, e_on2) and torch.equal(e_off, e_off2) and not torch.equal(e_on, e_off)

    def rel(a, b):
        return ((a - b).norm() / b.norm()).item()
    # fp32 atomics in the weight gradients: equal to a few 1e-7, not bit for bit
    assert rel(g_on_then_off, g_on) < 1e-5 and rel(g_off_then_on, g_off) < 1e-5
    assert rel(g_on, g_off) > 1e-2 and torch.isfinite(g_off_then_on).all()


def test_trainer_steps_with_dropout_and_eval_mode_is_unaffected():
    from quadruplet_sentence_transformer_amd.trainer import QuadrupletTrainer
    cfg = PRESETS["tiny-bert"]
    tr = QuadrupletTrainer(cfg, arena=synthetic_params(cfg, seed=14, std=0.05), device="cuda:0", lr=2e-3, dropout=0.1, dropout_seed=3)
    batch = [torch.from_numpy(x).cuda() for x in synthetic_quadruplets(cfg, 8, 32, seed=14)]
    l_eval0 = tr.forward_loss(*batch)[0].item()
    losses = [tr.step(*batch).item() for _ in range(12)]
    assert all(math.isfinite(v) for v in losses)
    assert len(set(round(v, 6) for v in losses)) > 6                 # the mask changes from step to step
    assert tr.forward_loss(*batch)[0].item() < l_eval0                # and the model still learns the batch
    assert tr.enc.dropout_step == 12
    with pytest.raises(ValueError):
        tr.enc.set_dropout(1.0, 0.0)


def test_captured_step_draws_fresh_masks_on_every_replay():
    """QuadrupletTrainer(use_graph=True): the counter advance and the per-forward snapshot are launches inside the captured
    step, so every replay uses the next step's masks -- the losses of the graph-replayed run follow those of the eager run
    with the same seed, step for step."""
    from quadruplet_sentence_transformer_amd.trainer import QuadrupletTrainer
    cfg = PRESETS["tiny-bert"]
    arena = synthetic_params(cfg, seed=14, std=0.05)
    batches = [[torch.from_numpy(x).cuda() for x in synthetic_quadruplets(cfg, 8, 32, seed=14, step=i)] for i in range(3)]
    runs = []
    for use_graph in (False, True):
        tr = QuadrupletTrainer(cfg, arena=arena, device="cuda:0", lr=1e-3, dropout=0.1, dropout_seed=9, use_graph=use_graph)
        runs.append([tr.step(*batches[i % 3]).item() for i in range(7)])
        torch.cuda.synchronize()
        assert tr.enc.drop_state.cpu().tolist()[2] == 7
    eager, graph = runs
    assert len(set(round(v, 6) for v in graph)) == 7
    np.testing.assert_allclose(graph, eager, rtol=0, atol=2e-4)
