"""GPU parity of every HIP kernel (through the C-ABI) against the CPU oracle / plain torch fp32.

Tolerances: kernels that round operands to bf16 are compared with a reference fed the SAME
bf16-rounded operands (fp32 accumulate), so only accumulation order differs: rtol 2e-3 on
bf16 outputs (one bf16 ulp = 2^-8 relative), 1e-4 on fp32 outputs.
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import quadruplet_sentence_transformer_amd as qst  # noqa: E402
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402
from quadruplet_sentence_transformer_amd.encoder import quadruplet_loss_raw  # noqa: E402
from oracle import torch_ref as R  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return _lib.load()


def dev(t):
    return t.cuda().contiguous()


def stream():
    return _lib.current_stream_ptr()


def bfr(t):
    return t.to(torch.bfloat16).to(torch.float32)


# The kernels with 16-bit matrix-core operands exist on bf16 (qst_*) and on IEEE half (qst_*_f16: QST_PREC_F16, the same
# sources compiled on the other operand type); their tests run on both.
OPDT = {"bf16": torch.bfloat16, "f16": torch.float16}


@pytest.fixture(params=["bf16", "f16"])
def op(request):
    return request.param


def opr(op, t):
    """t rounded to the operand type and back (what the kernel's operand holds)."""
    return t.to(OPDT[op]).to(torch.float32)


def kf(lib, name, op):
    return _lib.kfn(lib, name, op)


# ------------------------------------------------------------------ loss
@pytest.mark.parametrize("B,D", [(1, 10), (5, 10), (8, 384), (64, 384), (32, 768), (7, 33), (3, 2052)])
@pytest.mark.parametrize("p", [2.0, 1.0, 3.0])
@pytest.mark.parametrize("swap", [False, True])
def test_loss_matches_oracle(lib, B, D, p, swap):
    g = torch.Generator().manual_seed(B * 1000 + D)
    x = [torch.randn(B, D, generator=g) for _ in range(4)]
    if D == 384:
        x = [t / t.norm(dim=1, keepdim=True) for t in x]
    kw = dict(gamma=0.6, margin_pos_neg=1.0, margin_pos_part=0.5, margin_part_neg=0.5, p=p, swap=swap)
    for red_name, red in (("none", 0), ("sum", 1), ("mean", 2)):
        xs = [t.clone().requires_grad_(True) for t in x]
        ref = R.gamma_quadruplet_loss_ref(*xs, reduction=red_name, **kw)
        w = torch.linspace(0.5, 1.5, B) if red == 0 else torch.tensor([1.7])
        (ref * w).sum().backward()
        out, grads = quadruplet_loss_raw(*[dev(t) for t in x], 0.6, 1.0, 0.5, 0.5, p, swap, red,
                                         grad_out=dev(w), want_grads=True)
        # L1 / L3 distances of 384-dim rows are O(20): fp32 summation order alone moves them by ~2e-5; L2 distances of
        # un-normalised rows grow as sqrt(2 D) (64 at D = 2,052: one fp32 ulp there is 4e-6, tools/fuzz_shapes.py loss case 52)
        tol = max(1e-5, 1.5e-8 * D) if p == 2.0 else max(1e-4, 2e-6 * D)
        torch.testing.assert_close(out.cpu().view(ref.shape), ref.detach(), rtol=tol, atol=tol * (B if red == 1 else 1))
        for gi, xi in zip(grads, xs):
            torch.testing.assert_close(gi.cpu(), xi.grad, rtol=1e-4, atol=1e-6)


def test_loss_edge_cases(lib):
    B, D = 6, 64
    a = torch.randn(B, D)
    # all hinges inactive: positives at the anchor, negatives far away -> zero loss, zero grads
    far = a + 100.0
    out, grads = quadruplet_loss_raw(dev(a), dev(a + 1e-3), dev(a + 2e-3), dev(far), 0.6, 1.0, 0.5, 0.5, 2.0, False, 2,
                                     want_grads=True)
    # c = max(0.5 + d(a,p) - d(a,q), 0) is active here; use explicit reference
    xs = [t.clone().requires_grad_(True) for t in (a, a + 1e-3, a + 2e-3, far)]
    ref = R.gamma_quadruplet_loss_ref(*xs)
    ref.backward()
    torch.testing.assert_close(out.cpu()[0], ref.detach(), rtol=1e-5, atol=1e-6)
    for gi, xi in zip(grads, xs):
        torch.testing.assert_close(gi.cpu(), xi.grad, rtol=1e-4, atol=1e-7)
    # anchor == positive: distance is ||1e-6 * 1||
    out, _ = quadruplet_loss_raw(dev(a), dev(a), dev(a), dev(a), 0.6, 1.0, 0.5, 0.5, 2.0, False, 0)
    ref = R.gamma_quadruplet_loss_ref(a, a, a, a, reduction="none")
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-6, atol=1e-6)


def test_loss_bad_args(lib):
    x = torch.zeros(4, 8, device="cuda")
    s = torch.zeros(4, device="cuda")
    o = torch.zeros(4, device="cuda")
    rc = lib.qst_quadruplet_loss(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), 4, 8, 1.5, 1.0, 0.5, 0.5, 2.0,
                                 0, 2, o.data_ptr(), None, None, None, None, None, s.data_ptr(), stream())
    assert rc == -1


# ------------------------------------------------------------------ GEMM
def gemm_args(**kw):
    g = _lib.QstGemmArgs()
    g._keep = [v for v in kw.values() if torch.is_tensor(v)]   # keep device tensors alive until the launch is enqueued
    for k, v in kw.items():
        setattr(g, k, v.data_ptr() if torch.is_tensor(v) else v)
    return g


@pytest.mark.parametrize("form", [0, 2, 4, 0x20, 0x40], ids=["tiles128", "tiles256", "tall256", "eightphase128x384", "eightphase256x256"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 384), (200, 192, 128), (1000, 1152, 384), (64, 64, 64),
                                   (4096, 384, 1536), (5000, 1536, 384), (20000, 1152, 384), (12000, 768, 768),
                                   (33000, 384, 448)])
def test_gemm_nt_epilogues(lib, op, M, N, K, form):
    """form: QstGemmArgs.splits selects the nt tiling (0 = automatic; 1 = 128-row tiles, two workgroups per CU; 2 = 256-row
    tiles, one 8-wave workgroup per CU; 4 = 256-row tiles of four waves owning 128 x 96 each, bf16-output epilogues only --
    the fp32 one falls back to form 1; 0x20 / 0x40 = the 8-wave, 8-phase K loop of csrc/gemm8.hip with its 128 x 384 / 256 x 256
    workgroup tile, register-only epilogues -- K = 448 is an odd number of K-tiles, M = 33000 / 5000 / 200 ragged row tiles);
    all must give the same results."""
    g = torch.Generator().manual_seed(M + N + K)
    A = opr(op, torch.randn(M, K, generator=g))
    B = opr(op, torch.randn(N, K, generator=g) * 0.05)
    bias = torch.randn(N, generator=g)
    resid = torch.randn(M, N, generator=g)
    ref = A @ B.t() + bias
    Ad, Bd = dev(A.to(OPDT[op])), dev(B.to(OPDT[op]))
    # EPI_BF16
    Cb = torch.empty(M, N, dtype=OPDT[op], device="cuda")
    _lib.check(kf(lib, "qst_gemm_nt", op)(gemm_args(A=Ad, B=Bd, C=Cb, bias=dev(bias), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, splits=form), 0, stream()))
    torch.testing.assert_close(Cb.float().cpu(), ref, rtol=8e-3, atol=2e-2)
    # EPI_F32_RESID
    Cf = torch.empty(M, N, dtype=torch.float32, device="cuda")
    _lib.check(kf(lib, "qst_gemm_nt", op)(gemm_args(A=Ad, B=Bd, C=Cf, bias=dev(bias), resid=dev(resid), M=M, N=N, K=K, lda=K, ldb=K,
                                         ldc=N, ldr=N, splits=form), 1, stream()))
    torch.testing.assert_close(Cf.cpu(), ref + resid, rtol=1e-4, atol=1e-3)
    # EPI_GELU: C = gelu'(u) (saved for backward), C2 = gelu(u)
    C2 = torch.empty(M, N, dtype=OPDT[op], device="cuda")
    _lib.check(kf(lib, "qst_gemm_nt", op)(gemm_args(A=Ad, B=Bd, C=Cb, C2=C2, bias=dev(bias), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, splits=form), 2, stream()))
    ur = ref.clone().requires_grad_(True)
    torch.nn.functional.gelu(ur).sum().backward()
    torch.testing.assert_close(Cb.float().cpu(), ur.grad, rtol=8e-3, atol=2e-2)
    torch.testing.assert_close(C2.float().cpu(), torch.nn.functional.gelu(ref), rtol=8e-3, atol=2e-2)
    # EPI_GELU_BWD: C = acc * aux
    gp = opr(op, torch.rand(M, N, generator=g) * 1.2 - 0.1)
    _lib.check(kf(lib, "qst_gemm_nt", op)(gemm_args(A=Ad, B=Bd, C=Cb, aux=dev(gp.to(OPDT[op])), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, splits=form), 3, stream()))
    torch.testing.assert_close(Cb.float().cpu(), (A @ B.t()) * gp, rtol=8e-3, atol=2e-2)


@pytest.fixture(params=[0, 2], ids=["tiled", "eightphase"])
def tn_mode(request, lib):
    """qst_gemm8_mode: 0 = the tiled weight-gradient kernel of csrc/gemm.hip, 2 = the 8-phase one of csrc/gemm8.hip"""
    lib.qst_gemm8_mode(request.param)
    yield request.param
    lib.qst_gemm8_mode(-1)


@pytest.mark.parametrize("M,N,K", [(64, 128, 128), (512, 384, 384), (1000, 1152, 384), (300, 64, 256), (4096, 384, 1536),
                                   (96, 192, 64), (16384, 384, 384), (32768, 768, 768), (8256, 192, 384), (777, 200, 136)])
def test_gemm_tn_wgrad(lib, op, tn_mode, M, N, K):
    """Single-problem weight gradient. The large cases have fewer tiles than workgroups per M-range (every tile is cut into
    stage pieces: VERDICT r02 weak point 9 -- tools/gemm_bench.py died on (32768, 768, 768) with no test at that size) and an
    M that leaves a ragged last 64-row stage."""
    g = torch.Generator().manual_seed(M * 3 + N + K)
    A = opr(op, torch.randn(M, N, generator=g))      # dY
    B = opr(op, torch.randn(M, K, generator=g))      # X
    ref = A.t() @ B
    C = torch.zeros(N, K, dtype=torch.float32, device="cuda")
    cs = torch.zeros(N, dtype=torch.float32, device="cuda")
    _lib.check(kf(lib, "qst_gemm_tn", op)(gemm_args(A=dev(A.to(OPDT[op])), B=dev(B.to(OPDT[op])), C=C, colsum=cs, M=M, N=N,
                                         K=K, lda=N, ldb=K, ldc=K, splits=0), stream()))
    scale = math.sqrt(M)
    torch.testing.assert_close(C.cpu(), ref, rtol=1e-3, atol=1e-3 * scale)
    torch.testing.assert_close(cs.cpu(), A.sum(0), rtol=1e-3, atol=1e-3 * scale)
    # accumulation semantics: a second call doubles the result
    _lib.check(kf(lib, "qst_gemm_tn", op)(gemm_args(A=dev(A.to(OPDT[op])), B=dev(B.to(OPDT[op])), C=C, colsum=cs, M=M, N=N,
                                         K=K, lda=N, ldb=K, ldc=K, splits=3), stream()))
    torch.testing.assert_close(C.cpu(), 2 * ref, rtol=1e-3, atol=2e-3 * scale)


@pytest.mark.parametrize("M", [1000, 4096, 32 * 70 + 32, 96])
def test_gemm_tn_group_matches_individual(lib, op, tn_mode, M):
    """All four weight gradients of a MiniLM-shaped layer in one grouped launch -- incl. a last M-range shorter than
    the others and M too small for eight ranges."""
    H, I = 384, 1536
    g = torch.Generator().manual_seed(5 + M)
    shapes = [(H, I), (I, H), (H, H), (3 * H, H)]
    grp = _lib.QstTnGroup()
    grp.nprob, grp.splits = 4, 0
    keep, refs, outs = [], [], []
    for i, (N, K) in enumerate(shapes):
        A = opr(op, torch.randn(M, N, generator=g))
        B = opr(op, torch.randn(M, K, generator=g))
        Ad, Bd = dev(A.to(OPDT[op])), dev(B.to(OPDT[op]))
        C = torch.ones(N, K, device="cuda")                    # accumulation semantics: C += A^T.B
        cs = torch.zeros(N, device="cuda")
        q = grp.prob[i]
        q.A, q.B, q.C, q.colsum = Ad.data_ptr(), Bd.data_ptr(), C.data_ptr(), cs.data_ptr()
        q.M, q.N, q.K, q.lda, q.ldb, q.ldc = M, N, K, N, K, K
        keep += [Ad, Bd]
        refs.append((A.t() @ B + 1.0, A.sum(0)))
        outs.append((C, cs))
    _lib.check(kf(lib, "qst_gemm_tn_group", op)(grp, stream()))
    torch.cuda.synchronize()
    for (C, cs), (rC, rcs) in zip(outs, refs):
        torch.testing.assert_close(C.cpu(), rC, rtol=1e-3, atol=1e-3 * math.sqrt(M))
        torch.testing.assert_close(cs.cpu(), rcs, rtol=1e-3, atol=1e-3 * math.sqrt(M))


# ------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("M,H", [(37, 64), (128, 128), (300, 384), (129, 768), (5, 1024)])
def test_layernorm_fwd_bwd(lib, op, M, H):
    g = torch.Generator().manual_seed(M + H)
    s = torch.randn(M, H, generator=g) * 2 + 0.3
    gamma = 1 + 0.1 * torch.randn(H, generator=g)
    beta = 0.1 * torch.randn(H, generator=g)
    dy = torch.randn(M, H, generator=g)
    sr = s.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    yref = torch.nn.functional.layer_norm(sr, (H,), gr, br, 1e-12)
    (yref * dy).sum().backward()
    y = torch.empty(M, H, device="cuda")
    yb = torch.empty(M, H, dtype=OPDT[op], device="cuda")
    xh = torch.empty(M, H, dtype=OPDT[op], device="cuda")
    rs = torch.empty(M, device="cuda")
    sd, gd, bd, dyd = dev(s), dev(gamma), dev(beta), dev(dy)      # keep the device tensors alive across the launches
    _lib.check(kf(lib, "qst_ln_fwd", op)(sd.data_ptr(), gd.data_ptr(), bd.data_ptr(), 1e-12, M, H, y.data_ptr(),
                              yb.data_ptr(), xh.data_ptr(), rs.data_ptr(), stream()))
    torch.testing.assert_close(y.cpu(), yref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(yb.float().cpu(), yref.detach(), rtol=8e-3, atol=1e-2)
    ds = torch.empty(M, H, device="cuda")
    dsb = torch.empty(M, H, dtype=OPDT[op], device="cuda")
    dg = torch.zeros(H, device="cuda")
    db = torch.zeros(H, device="cuda")
    scratch = torch.empty(lib.qst_ln_bwd_scratch_bytes(M, H) // 4, device="cuda")
    _lib.check(kf(lib, "qst_ln_bwd", op)(dyd.data_ptr(), xh.data_ptr(), rs.data_ptr(), gd.data_ptr(), M, H, ds.data_ptr(),
                              dsb.data_ptr(), dg.data_ptr(), db.data_ptr(), scratch.data_ptr(), stream()))
    dg2 = torch.zeros(H, device="cuda")
    db2 = torch.zeros(H, device="cuda")
    _lib.check(kf(lib, "qst_ln_bwd", op)(dyd.data_ptr(), xh.data_ptr(), rs.data_ptr(), gd.data_ptr(), M, H, ds.data_ptr(),
                              dsb.data_ptr(), dg2.data_ptr(), db2.data_ptr(), None, stream()))      # atomic fallback
    torch.testing.assert_close(dg2, dg, rtol=1e-4, atol=1e-4 * math.sqrt(M))
    torch.testing.assert_close(db2, db, rtol=1e-4, atol=1e-4 * math.sqrt(M))
    # xhat is stored in bf16 -> 2^-9 relative perturbation of the xhat terms
    torch.testing.assert_close(ds.cpu(), sr.grad, rtol=2e-2, atol=2e-2 * sr.grad.abs().max().item())
    torch.testing.assert_close(db.cpu(), br.grad, rtol=1e-4, atol=1e-4 * math.sqrt(M))
    torch.testing.assert_close(dg.cpu(), gr.grad, rtol=1e-2, atol=1e-2 * math.sqrt(M))


@pytest.mark.parametrize("M,K,N", [(128, 64, 384), (300, 384, 384), (1000, 1536, 384), (4096, 1152, 384),
                                   (300, 768, 768), (1000, 3072, 768), (4096, 2304, 768), (8200, 768, 768), (600, 128, 512),
                                   (257, 64, 1024), (100, 128, 768), (1, 64, 512),
                                   (33000, 768, 768), (40001, 128, 768)])          # the 128 x 384 tile (N = 768, 32,768 <= M < 43,691)
def test_gemm_nt_fused_layernorm(lib, op, M, K, N):
    """qst_gemm_nt_ln (N = 384: full-row tiles; N = 512 / 768 / 1024: the workgroups of a row panel exchange the row
    statistics inside the launch, gemm8.hip) against the unfused pair it replaces: qst_gemm_nt(F32_RESID) followed
    by qst_ln_fwd / qst_ln_bwd -- same arithmetic, so fp32 outputs agree to accumulation-order noise."""
    assert lib.qst_gemm_nt_ln_supported(N) == 1 and lib.qst_gemm_nt_ln_supported(192) == 0
    g = torch.Generator().manual_seed(M + K)
    Ad = dev(opr(op, torch.randn(M, K, generator=g)).to(OPDT[op]))
    Bd = dev(opr(op, torch.randn(N, K, generator=g) * 0.05).to(OPDT[op]))
    bias, resid = dev(torch.randn(N, generator=g)), dev(torch.randn(M, N, generator=g))
    gamma, beta = dev(1 + 0.1 * torch.randn(N, generator=g)), dev(0.1 * torch.randn(N, generator=g))
    eps = 1e-12

    def ln_epi(**kw):
        e = _lib.QstLnEpi()
        e._keep = [v for v in kw.values() if torch.is_tensor(v)]
        for k, v in kw.items():
            setattr(e, k, v.data_ptr() if torch.is_tensor(v) else v)
        return e

    def f32(*shape):
        return torch.empty(*shape, dtype=torch.float32, device="cuda")

    def b16(*shape):
        return torch.empty(*shape, dtype=OPDT[op], device="cuda")

    # ---- forward: y = LN(A.B^T + bias + resid)
    s = f32(M, N)
    _lib.check(kf(lib, "qst_gemm_nt", op)(gemm_args(A=Ad, B=Bd, C=s, bias=bias, resid=resid, M=M, N=N, K=K, lda=K, ldb=K, ldc=N,
                                         ldr=N), 1, stream()))
    y0, yb0, xh0, rs0 = f32(M, N), b16(M, N), b16(M, N), f32(M)
    _lib.check(kf(lib, "qst_ln_fwd", op)(s.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps, M, N, y0.data_ptr(), yb0.data_ptr(),
                              xh0.data_ptr(), rs0.data_ptr(), stream()))
    y1, yb1, xh1, rs1 = f32(M, N), b16(M, N), b16(M, N), f32(M)
    _lib.check(kf(lib, "qst_gemm_nt_ln", op)(gemm_args(A=Ad, B=Bd, C=y1, C2=yb1, bias=bias, resid=resid, M=M, N=N, K=K, lda=K, ldb=K,
                                            ldc=N, ldr=N),
                                  ln_epi(gamma=gamma, beta=beta, eps=eps, xhat=xh1, rstd=rs1), 0, stream()))
    torch.testing.assert_close(y1, y0, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(rs1, rs0, rtol=1e-5, atol=0)
    torch.testing.assert_close(yb1.float(), yb0.float(), rtol=8e-3, atol=1e-2)
    torch.testing.assert_close(xh1.float(), xh0.float(), rtol=8e-3, atol=1e-2)
    ref = torch.nn.functional.layer_norm(Ad.float() @ Bd.float().t() + bias + resid, (N,), gamma, beta, eps)
    torch.testing.assert_close(y1, ref, rtol=1e-4, atol=2e-4)
    # outputs that the caller does not want may be NULL (inference)
    y2 = f32(M, N)
    _lib.check(kf(lib, "qst_gemm_nt_ln", op)(gemm_args(A=Ad, B=Bd, C=y2, bias=bias, resid=resid, M=M, N=N, K=K, lda=K, ldb=K, ldc=N,
                                            ldr=N), ln_epi(gamma=gamma, beta=beta, eps=eps), 0, stream()))
    torch.testing.assert_close(y2, y1, rtol=0, atol=0)

    # ---- backward: ds = LN_bwd(A.B^T + resid), dgamma/dbeta through per-tile partial sums
    dy = f32(M, N)
    _lib.check(kf(lib, "qst_gemm_nt", op)(gemm_args(A=Ad, B=Bd, C=dy, resid=resid, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, ldr=N), 1,
                               stream()))
    ds0, dsb0 = f32(M, N), b16(M, N)
    dg0, db0 = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
    scratch = torch.empty(lib.qst_ln_bwd_scratch_bytes(M, N) // 4, device="cuda")
    _lib.check(kf(lib, "qst_ln_bwd", op)(dy.data_ptr(), xh0.data_ptr(), rs0.data_ptr(), gamma.data_ptr(), M, N, ds0.data_ptr(),
                              dsb0.data_ptr(), dg0.data_ptr(), db0.data_ptr(), scratch.data_ptr(), stream()))
    ds1, dsb1 = f32(M, N), b16(M, N)
    br = lib.qst_gemm_nt_ln_block_rows_m(N, M)
    ntile = (M + br - 1) // br
    part = torch.full((ntile, 2, N), float("nan"), device="cuda")
    _lib.check(kf(lib, "qst_gemm_nt_ln", op)(gemm_args(A=Ad, B=Bd, C=ds1, C2=dsb1, resid=resid, M=M, N=N, K=K, lda=K, ldb=K, ldc=N,
                                            ldr=N),
                                  ln_epi(gamma=gamma, xhat=xh0, rstd=rs0, partials=part), 1, stream()))
    scale = ds0.abs().max().item()
    torch.testing.assert_close(ds1, ds0, rtol=1e-4, atol=1e-4 * scale)
    torch.testing.assert_close(dsb1.float(), dsb0.float(), rtol=8e-3, atol=1e-2 * scale)
    torch.testing.assert_close(part[:, 0].sum(0), dg0, rtol=1e-4, atol=1e-4 * math.sqrt(M) * dy.abs().max().item())
    torch.testing.assert_close(part[:, 1].sum(0), db0, rtol=1e-4, atol=1e-4 * math.sqrt(M) * dy.abs().max().item())
    # bad shapes are refused, not mis-computed
    assert kf(lib, "qst_gemm_nt_ln", op)(gemm_args(A=Ad, B=Bd, C=ds1, M=M, N=192, K=K, lda=K, ldb=K, ldc=192),
                              ln_epi(gamma=gamma, xhat=xh0, rstd=rs0), 1, stream()) == -2
    assert kf(lib, "qst_gemm_nt8_ln_timeouts", op)() == 0        # no exchange of row statistics ever gave up waiting


def test_gemm_nt_fused_layernorm_n768_under_graph_capture(lib):
    """The several-tiles-per-row form tags its exchange granules with an epoch that lives on the device (the last workgroup
    of a launch advances it) and takes, while a stream is capturing, the buffer an eager launch has left: a captured launch
    replays to the eager result, three times, with eager launches in between."""
    M, K, N = 1300, 768, 768
    g = torch.Generator().manual_seed(5)
    A = dev(torch.randn(M, K, generator=g).to(torch.bfloat16)); B = dev((torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16))
    bias, resid = dev(torch.randn(N, generator=g)), dev(torch.randn(M, N, generator=g))
    gamma, beta = dev(1 + 0.1 * torch.randn(N, generator=g)), dev(0.1 * torch.randn(N, generator=g))
    y0, y1 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    e = _lib.QstLnEpi()
    e.gamma, e.beta, e.eps = gamma.data_ptr(), beta.data_ptr(), 1e-12

    def launch(out):
        _lib.check(lib.qst_gemm_nt_ln(gemm_args(A=A, B=B, C=out, bias=bias, resid=resid, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, ldr=N),
                                      e, 0, stream()))
    launch(y0)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        launch(y1)
    y2 = torch.empty(M, N, device="cuda")
    for _ in range(3):
        y1.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(y1, y0)
        launch(y2)
        torch.cuda.synchronize()
        assert torch.equal(y2, y0)
    assert lib.qst_gemm_nt8_ln_timeouts() == 0


@pytest.mark.parametrize("M,I", [(128, 192), (416, 768), (1000, 1536), (4096, 1536)])
def test_ffn_chain_matches_the_two_kernel_path(lib, op, M, I):
    """qst_ffn_chain (csrc/ffn.hip: the feed-forward block as one kernel) against the launches it replaces --
    forward: qst_gemm_nt(GELU) + qst_gemm_nt_ln(mode 0); backward: qst_gemm_nt(GELU_BWD) + qst_gemm_nt_ln(mode 1).
    The rounding points are the same (h / du are rounded to bf16 before the second product in both forms), so fp32
    outputs agree to accumulation-order noise; plus a plain fp32 torch reference of the forward."""
    H = 384
    assert lib.qst_ffn_chain_supported(H, I) == 1 and lib.qst_ffn_chain_supported(768, 3072) == 0
    g = torch.Generator().manual_seed(M + I)
    bf = OPDT[op]
    A = dev(opr(op, torch.randn(M, H, generator=g)).to(bf))
    W1 = dev(opr(op, torch.randn(I, H, generator=g) * 0.05).to(bf))
    W2 = dev(opr(op, torch.randn(H, I, generator=g) * 0.05).to(bf))
    W1t, W2t = W1.t().contiguous(), W2.t().contiguous()                 # the "W^T shadows" the dgrads read
    b1, b2 = dev(0.3 * torch.randn(I, generator=g)), dev(0.3 * torch.randn(H, generator=g))
    resid = dev(torch.randn(M, H, generator=g))
    gamma, beta = dev(1 + 0.1 * torch.randn(H, generator=g)), dev(0.1 * torch.randn(H, generator=g))
    eps = 1e-12

    def ln_epi(**kw):
        e = _lib.QstLnEpi()
        e._keep = [v for v in kw.values() if torch.is_tensor(v)]
        for k, v in kw.items():
            setattr(e, k, v.data_ptr() if torch.is_tensor(v) else v)
        return e

    def ffn_args(**kw):
        a = _lib.QstFfnArgs()
        a._keep = [v for v in kw.values() if torch.is_tensor(v)]
        for k, v in kw.items():
            setattr(a, k, v.data_ptr() if torch.is_tensor(v) else v)
        return a

    def f32(*shape):
        return torch.empty(*shape, dtype=torch.float32, device="cuda")

    def b16(*shape):
        return torch.empty(*shape, dtype=bf, device="cuda")

    # ---- forward, two kernels
    gp0, h0 = b16(M, I), b16(M, I)
    _lib.check(kf(lib, "qst_gemm_nt", op)(gemm_args(A=A, B=W1, C=gp0, C2=h0, bias=b1, M=M, N=I, K=H, lda=H, ldb=H, ldc=I), 2, stream()))
    y0, yb0, xh0, rs0 = f32(M, H), b16(M, H), b16(M, H), f32(M)
    _lib.check(kf(lib, "qst_gemm_nt_ln", op)(gemm_args(A=h0, B=W2, C=y0, C2=yb0, bias=b2, resid=resid, M=M, N=H, K=I, lda=I, ldb=I,
                                            ldc=H, ldr=H), ln_epi(gamma=gamma, beta=beta, eps=eps, xhat=xh0, rstd=rs0), 0,
                                  stream()))
    # ---- forward, one kernel (training: side outputs; inference: none)
    gp1, h1 = torch.full((M, I), 7.0, dtype=bf, device="cuda"), torch.full((M, I), 7.0, dtype=bf, device="cuda")
    y1, yb1, xh1, rs1 = f32(M, H), b16(M, H), b16(M, H), f32(M)
    _lib.check(kf(lib, "qst_ffn_chain", op)(ffn_args(A=A, B1=W1, B2=W2, bias1=b1, bias2=b2, resid=resid, save_gp=gp1, save_h=h1, C=y1,
                                          C2=yb1, M=M, H=H, I=I),
                                 ln_epi(gamma=gamma, beta=beta, eps=eps, xhat=xh1, rstd=rs1), 0, stream()))
    torch.cuda.synchronize()
    assert (gp1.float() - gp0.float()).abs().max().item() <= 2 ** -7 and (h1.float() - h0.float()).abs().max().item() <= \
        2 ** -7 * max(1.0, h0.float().abs().max().item())
    assert (h1 != h0).float().mean().item() < 1e-3                      # the same values up to rare last-bit roundings
    torch.testing.assert_close(y1, y0, rtol=1e-4, atol=2e-4)
    torch.testing.assert_close(rs1, rs0, rtol=1e-4, atol=0)
    torch.testing.assert_close(yb1.float(), yb0.float(), rtol=8e-3, atol=1e-2)
    torch.testing.assert_close(xh1.float(), xh0.float(), rtol=8e-3, atol=1e-2)
    u = A.float() @ W1.float().t() + b1
    href = opr(op, torch.nn.functional.gelu(u))
    ref = torch.nn.functional.layer_norm(href @ W2.float().t() + b2 + resid, (H,), gamma, beta, eps)
    torch.testing.assert_close(y1, ref, rtol=1e-3, atol=2e-3)
    y2 = f32(M, H)
    _lib.check(kf(lib, "qst_ffn_chain", op)(ffn_args(A=A, B1=W1, B2=W2, bias1=b1, bias2=b2, resid=resid, C=y2, M=M, H=H, I=I),
                                 ln_epi(gamma=gamma, beta=beta, eps=eps), 0, stream()))
    torch.testing.assert_close(y2, y1, rtol=0, atol=0)                  # inference variant: same arithmetic

    # ---- backward: du = (ds2 . W2) * gelu'(u) ; ds1 = LN1'(du . W1 + ds2)
    ds2 = dev(torch.randn(M, H, generator=g))
    ds2b = ds2.to(bf)
    du0 = b16(M, I)
    _lib.check(kf(lib, "qst_gemm_nt", op)(gemm_args(A=ds2b, B=W2t, C=du0, aux=gp0, M=M, N=I, K=H, lda=H, ldb=H, ldc=I), 3, stream()))
    ntile = (M + 127) // 128
    o0, ob0, part0 = f32(M, H), b16(M, H), f32(ntile, 2, H)
    _lib.check(kf(lib, "qst_gemm_nt_ln", op)(gemm_args(A=du0, B=W1t, C=o0, C2=ob0, resid=ds2, M=M, N=H, K=I, lda=I, ldb=I, ldc=H, ldr=H),
                                  ln_epi(gamma=gamma, xhat=xh0, rstd=rs0, partials=part0), 1, stream()))
    du1 = torch.full((M, I), 7.0, dtype=bf, device="cuda")
    o1, ob1, part1 = f32(M, H), b16(M, H), torch.full((ntile, 2, H), float("nan"), device="cuda")
    _lib.check(kf(lib, "qst_ffn_chain", op)(ffn_args(A=ds2b, B1=W2t, B2=W1t, resid=ds2, aux=gp0, save_h=du1, C=o1, C2=ob1, M=M, H=H, I=I),
                                 ln_epi(gamma=gamma, xhat=xh0, rstd=rs0, partials=part1), 1, stream()))
    torch.cuda.synchronize()
    sc = du0.float().abs().max().item()
    assert (du1.float() - du0.float()).abs().max().item() <= 2 ** -7 * sc and (du1 != du0).float().mean().item() < 1e-3
    scale = o0.abs().max().item()
    torch.testing.assert_close(o1, o0, rtol=1e-4, atol=2e-4 * scale)
    torch.testing.assert_close(ob1.float(), ob0.float(), rtol=8e-3, atol=1e-2 * scale)
    torch.testing.assert_close(part1.sum(0), part0.sum(0), rtol=1e-3, atol=1e-3 * math.sqrt(M) * scale)
    # refused, not mis-computed
    assert kf(lib, "qst_ffn_chain", op)(ffn_args(A=A, B1=W1, B2=W2, C=y2, M=M, H=768, I=I), ln_epi(gamma=gamma, beta=beta), 0, stream()) == -2
    assert kf(lib, "qst_ffn_chain", op)(ffn_args(A=A, B1=W1, B2=W2, C=y2, save_h=h1, M=M, H=H, I=I), ln_epi(gamma=gamma, beta=beta), 0,
                             stream()) == -1                            # save_h without save_gp


# ------------------------------------------------------------------ attention
def attn_ref(qkv, mask, rel, n, L, A, d):
    H = A * d
    q, k, v = [t.view(n, L, A, d).transpose(1, 2) for t in qkv.view(n, L, 3 * H).split(H, dim=-1)]
    s = q @ k.transpose(-1, -2) / math.sqrt(d)
    if rel is not None:
        s = s + rel[None]
    s = s + (1.0 - mask[:, None, None, :].float()) * torch.finfo(torch.float32).min
    p = torch.softmax(s, -1)
    return (p @ v).transpose(1, 2).reshape(n * L, H)


@pytest.mark.parametrize("n,L,A,d,use_rel", [(2, 32, 2, 32, False), (3, 128, 12, 32, False), (2, 160, 2, 64, True),
                                              (2, 256, 3, 64, False), (1, 384, 2, 64, True), (2, 64, 2, 32, True),
                                              (5, 96, 3, 32, False), (2, 160, 2, 32, True), (2, 512, 2, 64, False),
                                              (3, 288, 2, 64, True), (1, 512, 1, 64, True), (2, 32, 1, 64, False),
                                              (2, 384, 3, 64, False)])
def test_attention_fwd_bwd(lib, op, n, L, A, d, use_rel):
    H = A * d
    g = torch.Generator().manual_seed(n * L + A + d)
    qkv = opr(op, torch.randn(n * L, 3 * H, generator=g))
    lens = torch.randint(max(1, L // 8), L + 1, (n,), generator=g)
    lens[0] = L
    mask = (torch.arange(L)[None, :] < lens[:, None]).long()
    # MPNet's bias depends on j - i only: the kernels take it as relative-position vectors [A, 2L] (entry j - i + L)
    relpos = (0.5 * torch.randn(A, 2 * L, generator=g)) if use_rel else None
    ridx = (torch.arange(L)[None, :] - torch.arange(L)[:, None]) + L          # [i, j] -> j - i + L
    dctx = opr(op, torch.randn(n * L, H, generator=g))
    qr = qkv.clone().requires_grad_(True)
    relr = relpos.clone().requires_grad_(True) if use_rel else None
    ref = attn_ref(qr, mask, relr[:, ridx] if use_rel else None, n, L, A, d)
    (ref * dctx).sum().backward()

    qd = dev(qkv.to(OPDT[op]))
    md = dev(mask)
    reld = dev(relpos) if use_rel else None
    ctx = torch.empty(n * L, H, dtype=OPDT[op], device="cuda")
    lse = torch.empty(n, A, L, device="cuda")
    _lib.check(kf(lib, "qst_attention_fwd", op)(qd.data_ptr(), md.data_ptr(), _lib.ptr(reld), n, L, A, d, ctx.data_ptr(), lse.data_ptr(), stream()))
    torch.testing.assert_close(ctx.float().cpu(), ref.detach(), rtol=2e-2, atol=2e-2)

    dq = torch.empty(n * L, 3 * H, dtype=OPDT[op], device="cuda")
    drel = torch.zeros(A, 2 * L, device="cuda") if use_rel else None
    dcd = dev(dctx.to(OPDT[op]))
    delta = torch.empty(n, A, L, device="cuda")
    _lib.check(kf(lib, "qst_attention_bwd", op)(qd.data_ptr(), ctx.data_ptr(), dcd.data_ptr(), lse.data_ptr(),
                                     md.data_ptr(), _lib.ptr(reld), n, L, A, d, dq.data_ptr(), _lib.ptr(drel),
                                     delta.data_ptr(), stream()))
    gref = qr.grad
    err = (dq.float().cpu() - gref).abs().max().item()
    assert err <= 3e-2 * max(1.0, gref.abs().max().item()), f"dqkv max err {err}"
    rel_l2 = ((dq.float().cpu() - gref).norm() / gref.norm()).item()
    assert rel_l2 < 1e-2, f"dqkv relative L2 error {rel_l2}"
    if use_rel:
        rel_l2 = ((drel.cpu() - relr.grad).norm() / relr.grad.norm()).item()
        assert rel_l2 < 1e-2, f"drel relative L2 error {rel_l2}"
    if (L <= 128 and d == 32) or d == 64:
        # this shape ran a one-workgroup-per-(sequence, head) backward (d = 32, L <= 128: attn_bwd_fused_kernel; d = 64:
        # attn_bwd_one64_kernel, except with the position bias at L > 384, where its LDS does not fit); the two-kernel path
        # must agree with it
        dq2 = torch.empty_like(dq)
        drel2 = torch.zeros(A, 2 * L, device="cuda") if use_rel else None
        q = _lib.QstAttnDesc()
        q.qkv, q.mask, q.rel_pos, q.nseq, q.L, q.A, q.d = qd.data_ptr(), md.data_ptr(), _lib.ptr(reld), n, L, A, d
        q.ctx, q.lse, q.dctx, q.dqkv, q.drel = ctx.data_ptr(), lse.data_ptr(), dcd.data_ptr(), dq2.data_ptr(), _lib.ptr(drel2)
        # force_split = 1: the two-kernel path (the first call above ran the one-workgroup kernel where there is one)
        q.delta_scratch, q.force_split = delta.data_ptr(), 1
        _lib.check(kf(lib, "qst_attention_bwd_ex", op)(q, stream()))
        torch.cuda.synchronize()
        torch.testing.assert_close(dq.float(), dq2.float(), rtol=2e-2, atol=2e-2 * max(1.0, gref.abs().max().item()))
        if use_rel:
            torch.testing.assert_close(drel, drel2, rtol=1e-3, atol=1e-3 * max(1.0, drel2.abs().max().item()))
        rel_l2 = ((dq2.float().cpu() - gref).norm() / gref.norm()).item()
        assert rel_l2 < 1e-2, f"two-kernel backward: dqkv relative L2 error {rel_l2}"


@pytest.mark.parametrize("n,L,A,d,use_rel,drop", [(2, 32, 2, 32, False, False), (3, 128, 12, 32, False, True), (2, 160, 2, 64, True, False),
                                                   (2, 288, 2, 64, True, True), (1, 512, 2, 64, False, False), (2, 96, 3, 32, True, True),
                                                   (3, 384, 2, 64, False, True)])
def test_attention_bwd_x3_matches_the_fp32_kernel_and_autograd(lib, n, L, A, d, use_rel, drop):
    """The parity-precision attention backward (csrc/x3.hip: split-bf16 x3 MFMAs, fp32 softmax / dS) against the scalar fp32
    kernel it replaced in the bf16x3 backward (csrc/x3_bwd.hip, same dropout masks) and, without dropout, fp64 autograd of
    HF's attention (modeling_bert.py:111-136 / modeling_mpnet.py:149-158). One sequence is all padding but its first token."""
    H = A * d
    g = torch.Generator().manual_seed(7 * n * L + A + d)
    qkv = torch.randn(n * L, 3 * H, generator=g)
    lens = torch.randint(max(1, L // 8), L + 1, (n,), generator=g)
    lens[0] = L
    lens[-1] = 1 if n > 1 else L
    mask = (torch.arange(L)[None, :] < lens[:, None]).long()
    rel = (0.5 * torch.randn(A, L, L, generator=g)) if use_rel else None
    dctx = torch.randn(n * L, H, generator=g)
    qd, md, reld, dcd = dev(qkv), dev(mask), (dev(rel) if use_rel else None), dev(dctx)
    state = torch.tensor([14, 0, 3, 0], dtype=torch.int32, device="cuda")
    dd = _lib.QstDrop()
    if drop:
        dd.state, dd.site, dd.thr16 = state.data_ptr(), 5, 6554
    dp = C.byref(dd) if drop else None
    ctx = torch.empty(n * L, H, device="cuda")
    _lib.check(lib.qst_attention_fwd_x3_drop(qd.data_ptr(), md.data_ptr(), _lib.ptr(reld), n, L, A, d, ctx.data_ptr(), dp, stream()))
    out = {}
    for which in ("f32", "x3"):
        dq = torch.full((n * L, 3 * H), float("nan"), device="cuda")
        drel = torch.zeros(A, L, L, device="cuda") if use_rel else None
        if which == "f32":
            _lib.check(lib.qst_attention_bwd_f32_drop(qd.data_ptr(), ctx.data_ptr(), dcd.data_ptr(), md.data_ptr(), _lib.ptr(reld),
                                                      n, L, A, d, dq.data_ptr(), _lib.ptr(drel), dp, stream()))
        else:
            scratch = torch.empty(lib.qst_attention_bwd_x3_scratch_bytes(n, L, A) // 4, device="cuda")
            _lib.check(lib.qst_attention_bwd_x3(qd.data_ptr(), ctx.data_ptr(), dcd.data_ptr(), md.data_ptr(), _lib.ptr(reld),
                                                n, L, A, d, dq.data_ptr(), _lib.ptr(drel), scratch.data_ptr(), dp, stream()))
        torch.cuda.synchronize()
        out[which] = (dq.cpu(), drel.cpu() if use_rel else None)
    scale = out["f32"][0].abs().max().item()
    torch.testing.assert_close(out["x3"][0], out["f32"][0], rtol=1e-4, atol=2e-5 * max(1.0, scale))
    if use_rel:
        # (a query that sees one key has P = 1 and dS = dP - delta = 0 by cancellation: what is left is the rounding of the two
        # routes to dO . V -- 2^-17 per split-bf16 product against 2^-24 -- hence the absolute term)
        torch.testing.assert_close(out["x3"][1], out["f32"][1], rtol=1e-4, atol=1e-4 * max(1.0, out["f32"][1].abs().max().item()))
    if not drop:
        qr = qkv.double().requires_grad_(True)
        relr = rel.double().requires_grad_(True) if use_rel else None
        q, k, v = [t.view(n, L, A, d).transpose(1, 2) for t in qr.view(n, L, 3 * H).split(H, dim=-1)]
        sc = q @ k.transpose(-1, -2) / math.sqrt(d)
        if use_rel:
            sc = sc + relr[None]
        sc = sc + (1.0 - mask[:, None, None, :].double()) * torch.finfo(torch.float32).min
        ref = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(n * L, H)
        (ref * dctx.double()).sum().backward()
        torch.testing.assert_close(out["x3"][0].double(), qr.grad, rtol=1e-4, atol=2e-5 * max(1.0, scale))
        if use_rel:
            torch.testing.assert_close(out["x3"][1].double(), relr.grad, rtol=1e-4, atol=1e-4 * max(1.0, relr.grad.abs().max().item()))
    assert lib.qst_attention_bwd_x3(qd.data_ptr(), ctx.data_ptr(), dcd.data_ptr(), md.data_ptr(), None, n, L, A, d,
                                    out["x3"][0].data_ptr(), None, None, None, stream()) == -1           # no scratch
    assert lib.qst_attention_bwd_x3(qd.data_ptr(), ctx.data_ptr(), dcd.data_ptr(), md.data_ptr(), None, n, L, A, 48,
                                    out["x3"][0].data_ptr(), None, ctx.data_ptr(), None, stream()) == -2          # head size


@pytest.mark.parametrize("M,N,K", [(384, 384, 4096), (1152, 384, 32 * 70), (128, 1536, 8192), (100, 60, 64)])
def test_gemm_nt_x3_shared_reduction_and_column_sums(lib, M, N, K):
    """qst_gemm_nt_x3 form 3 (C += A B^T with the reduction shared among workgroups: the weight gradients of the bf16x3
    backward) and qst_colsum_f32 (its bias gradients) against fp64."""
    g = torch.Generator().manual_seed(M + N + K)
    A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    C0 = torch.randn(M, N, generator=g)
    Cd = dev(C0)
    Ad, Bd = dev(A), dev(B)
    _lib.check(lib.qst_gemm_nt_x3(gemm_args(A=Ad, B=Bd, C=Cd, M=M, N=N, K=K, lda=K, ldb=K, ldc=N), 3, stream()))
    want = C0.double() + A.double() @ B.double().t()
    torch.testing.assert_close(Cd.cpu().double(), want, rtol=1e-5, atol=2e-5 * math.sqrt(K))
    # the same product from operands stored with the reduction index as the row index (what the backward calls), + column sums
    At, Bt = dev(A.t().contiguous()), dev(B.t().contiguous())
    Cd2, cs0 = dev(C0), torch.randn(M, generator=g)
    csd = dev(cs0)
    for splits in (0, 64):
        Cd2.copy_(C0); csd.copy_(cs0)
        _lib.check(lib.qst_gemm_tn_x3(gemm_args(A=At, B=Bt, C=Cd2, colsum=csd, M=K, N=M, K=N, lda=M, ldb=N, ldc=N, splits=splits), stream()))
        torch.testing.assert_close(Cd2.cpu().double(), want, rtol=1e-5, atol=2e-5 * math.sqrt(K))
        torch.testing.assert_close(csd.cpu().double(), cs0.double() + A.double().sum(1), rtol=1e-5, atol=1e-5 * math.sqrt(K))
    x = torch.randn(K, N, generator=g)
    o0 = torch.randn(N, generator=g)
    od = dev(o0)
    _lib.check(lib.qst_colsum_f32(dev(x).data_ptr(), K, N, N, od.data_ptr(), stream()))
    torch.testing.assert_close(od.cpu().double(), o0.double() + x.double().sum(0), rtol=1e-5, atol=1e-5 * math.sqrt(K))


@pytest.mark.parametrize("nseq,L,H,vocab,ntypes,irregular_pos", [
    (5, 32, 64, 50, 2, False), (19, 32, 384, 3000, 2, False), (7, 96, 768, 200, 0, True), (33, 64, 128, 100000, 1, True),
    (3, 30, 64, 40, 2, False)])
def test_embedding_backward_scatter(lib, nseq, L, H, vocab, ntypes, irregular_pos):
    """qst_embed_bwd: ds rows added into the word / position / token-type tables in one pass, against index_add_; tables
    start non-zero (accumulation semantics); a heavily repeated id; MPNet-like position ids that differ from row to row; L
    not a multiple of 4, nseq not a multiple of 16."""
    g = torch.Generator().manual_seed(nseq * L + H)
    M = nseq * L
    ds = torch.randn(M, H, generator=g)
    ids = torch.randint(0, vocab, (M,), generator=g)
    ids[: M // 8] = ids[0]                                  # one heavily repeated id next to (for large vocabularies) unique ones
    types = torch.randint(0, max(1, ntypes), (M,), generator=g) if ntypes else None
    if irregular_pos:
        pos = (torch.arange(L)[None, :] + torch.randint(0, 3, (nseq, 1), generator=g)).reshape(-1).to(torch.int32)
    else:
        pos = torch.arange(L).repeat(nseq).to(torch.int32)
    npos = int(pos.max()) + 1
    w0, p0 = torch.randn(vocab, H, generator=g), torch.randn(npos, H, generator=g)
    t0 = torch.randn(max(1, ntypes), H, generator=g)
    want_w = w0.clone().index_add_(0, ids, ds)
    want_p = p0.clone().index_add_(0, pos.long(), ds)
    want_t = t0.clone().index_add_(0, types, ds) if ntypes else None
    dw, dp, dt = dev(w0), dev(p0), dev(t0)
    dsd, idd, posd, typd = dev(ds), dev(ids), dev(pos), (dev(types) if ntypes else None)
    _lib.check(lib.qst_embed_bwd(dsd.data_ptr(), idd.data_ptr(), _lib.ptr(typd), posd.data_ptr(), nseq, L, H, ntypes,
                                 dw.data_ptr(), dp.data_ptr(),
                                 dt.data_ptr() if ntypes else None, stream()))
    torch.cuda.synchronize()
    tol = dict(rtol=1e-5, atol=1e-4 * math.sqrt(M / 8))
    torch.testing.assert_close(dw.cpu(), want_w, **tol)
    torch.testing.assert_close(dp.cpu(), want_p, **tol)
    if ntypes:
        torch.testing.assert_close(dt.cpu(), want_t, **tol)


def test_rel_pos_vectors_match_the_full_bias_table(lib):
    """qst_rel_pos_fwd / qst_rel_pos_bwd (relative-position form used by the bf16 attention kernels) against
    qst_rel_bias_fwd / qst_rel_bias_bwd (the [A, L, L] table of the parity path): same bias, same table gradient."""
    A, L, buckets, maxd = 12, 96, 32, 128
    lut = torch.tensor([lib.qst_rel_bucket_host(r, buckets, maxd) for r in range(-511, 512)], dtype=torch.int32).cuda()
    table = torch.randn(buckets, A).cuda()
    full = torch.empty(A, L, L, device="cuda")
    vec = torch.empty(A, 2 * L, device="cuda")
    _lib.check(lib.qst_rel_bias_fwd(table.data_ptr(), lut.data_ptr(), A, L, full.data_ptr(), stream()))
    _lib.check(lib.qst_rel_pos_fwd(table.data_ptr(), lut.data_ptr(), A, L, vec.data_ptr(), stream()))
    ridx = ((torch.arange(L)[None, :] - torch.arange(L)[:, None]) + L).cuda()
    torch.testing.assert_close(vec[:, ridx], full, rtol=0, atol=0)
    assert float(vec[:, 0].abs().max()) == 0.0
    # gradient: a random d(full); its relative-position sums must reduce to the same table gradient
    dfull = torch.randn(A, L, L, device="cuda")
    dvec = torch.zeros(A, 2 * L, device="cuda")
    dvec.index_put_((torch.arange(A, device="cuda")[:, None, None].expand(A, L, L), ridx[None].expand(A, L, L)), dfull,
                    accumulate=True)
    dt_full, dt_vec = torch.zeros(buckets, A, device="cuda"), torch.zeros(buckets, A, device="cuda")
    _lib.check(lib.qst_rel_bias_bwd(dfull.data_ptr(), lut.data_ptr(), buckets, A, L, dt_full.data_ptr(), stream()))
    _lib.check(lib.qst_rel_pos_bwd(dvec.data_ptr(), lut.data_ptr(), buckets, A, L, dt_vec.data_ptr(), stream()))
    torch.testing.assert_close(dt_vec, dt_full, rtol=1e-4, atol=1e-3)


# ------------------------------------------------------------------ AdamW
def test_clip_adamw_matches_torch(lib):
    from quadruplet_sentence_transformer_amd.encoder import HipEncoder
    from quadruplet_sentence_transformer_amd.config import PRESETS
    from quadruplet_sentence_transformer_amd.synthetic import synthetic_params
    cfg = PRESETS["tiny-bert"]
    enc = HipEncoder(cfg)
    arena = synthetic_params(cfg, seed=3, std=0.05, bias_std=0.02, ln_jitter=0.05)
    enc.load_arena(arena)
    enc.ensure_train_state()
    # torch reference with ST's two parameter groups
    views = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in enc.named_views().items()}
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    groups = [{"params": [p for n, p in views.items() if not any(nd in n for nd in no_decay)], "weight_decay": 0.01},
              {"params": [p for n, p in views.items() if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
    opt = torch.optim.AdamW(groups, lr=1e-2)
    gen = torch.Generator().manual_seed(0)
    for step in range(3):
        gv = enc.grad_views()
        for n_, p in views.items():
            gr = torch.randn(p.shape, generator=gen) * (3.0 if step == 0 else 0.01)
            p.grad = gr.clone()
            gv[n_].copy_(gr)
        total = torch.nn.utils.clip_grad_norm_(list(views.values()), 1.0)
        opt.step()
        enc.adamw_step(lr=1e-2, weight_decay=0.01, max_grad_norm=1.0)
        torch.testing.assert_close(enc.grad_norm.cpu()[0], total, rtol=1e-4, atol=1e-6)
        for n_, p in views.items():
            torch.testing.assert_close(enc.named_views()[n_].cpu(), p.detach(), rtol=1e-5, atol=1e-6)
        assert float(enc.grads.abs().max()) == 0.0   # zero_grad fused
