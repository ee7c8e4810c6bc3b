#!/usr/bin/env python3
"""H = 384 (the headline shape): gemm_nt_ln_kernel (gemm.hip: 128 x 384 full-row tile, its own K loop) against the same
decomposition on the 8-phase loop (gemm8.hip gemm_nt8_ln_kernel<..., 4, 6> with one tile per row panel), M = 32,768."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402
from gemm_bench import timeit  # noqa: E402

lib = _lib.load()
st = _lib.current_stream_ptr()
bf = torch.bfloat16
M, N = 32768, 384


def gargs(**kw):
    g = _lib.QstGemmArgs()
    for k, v in kw.items():
        setattr(g, k, v.data_ptr() if torch.is_tensor(v) else v)
    return g


gamma = 1 + 0.1 * torch.randn(N, device="cuda"); beta = 0.1 * torch.randn(N, device="cuda"); bias = torch.randn(N, device="cuda")
resid = torch.randn(M, N, device="cuda")
y = [torch.empty(M, N, device="cuda") for _ in range(2)]; yb = [torch.empty(M, N, device="cuda", dtype=bf) for _ in range(2)]
xh = [torch.empty(M, N, device="cuda", dtype=bf) for _ in range(2)]; rs = [torch.empty(M, device="cuda") for _ in range(2)]
part = torch.zeros(M // 128, 2, N, device="cuda")
for K in (384, 1536, 1152):
    A = torch.randn(M, K, device="cuda").to(bf); B = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
    e = []
    for i in range(2):
        q = _lib.QstLnEpi(); q.gamma, q.beta, q.eps, q.xhat, q.rstd = gamma.data_ptr(), beta.data_ptr(), 1e-12, xh[i].data_ptr(), rs[i].data_ptr()
        e.append(q)
    f0 = lambda: _lib.check(lib.qst_gemm_nt_ln(gargs(A=A, B=B, C=y[0], C2=yb[0], bias=bias, resid=resid, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, ldr=N), e[0], 0, st))
    f1 = lambda: _lib.check(lib.qst_gemm_nt8_ln(gargs(A=A, B=B, C=y[1], C2=yb[1], bias=bias, resid=resid, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, ldr=N), e[1], 0, st))
    f0(); f1(); torch.cuda.synchronize()
    d = (y[0] - y[1]).abs().max().item()
    e1 = _lib.QstLnEpi(); e1.gamma, e1.xhat, e1.rstd, e1.partials = gamma.data_ptr(), xh[0].data_ptr(), rs[0].data_ptr(), part.data_ptr()
    b0 = lambda: _lib.check(lib.qst_gemm_nt_ln(gargs(A=A, B=B, C=y[0], C2=yb[0], resid=resid, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, ldr=N), e1, 1, st))
    b1 = lambda: _lib.check(lib.qst_gemm_nt8_ln(gargs(A=A, B=B, C=y[1], C2=yb[1], resid=resid, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, ldr=N), e1, 1, st))
    b0(); b1(); torch.cuda.synchronize()
    db = (y[0] - y[1]).abs().max().item() / y[0].abs().max().item()
    best = {}
    for _ in range(4):
        for name, fn in (("f0", f0), ("f1", f1), ("b0", b0), ("b1", b1)):
            best[name] = min(best.get(name, 1e9), timeit(fn, reps=20))
    print(f"K={K:5d}: forward gemm_nt_ln {best['f0']:6.1f} us, 8-phase {best['f1']:6.1f} ({best['f1'] / best['f0'] - 1:+.1%}) | backward {best['b0']:6.1f} us, "
          f"8-phase {best['b1']:6.1f} ({best['b1'] / best['b0'] - 1:+.1%}) | max|d| fwd {d:.1e}, bwd rel {db:.1e}", flush=True)
