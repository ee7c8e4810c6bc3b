#!/usr/bin/env python3
"""Staggered start of the one-workgroup-per-CU 8-phase NT launches (qst_gemm8_stagger): the launches of the configs[4] layer
at M = 196,608 for several spreads of the first round's start. usage: stagger_bench.py [M] [cycles ...]"""
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
H, I = 768, 3072


def gargs(**kw):
    g = _lib.QstGemmArgs()
    for k, v in kw.items():
        setattr(g, k, v.data_ptr() if torch.is_tensor(v) else v)
    return g


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 196608
    spreads = [int(a) for a in sys.argv[2:]] or [0, 20000, 40000, 80000, 160000]
    x = torch.randn(M, H, device="cuda").to(bf); xi = torch.randn(M, I, device="cuda").to(bf)
    W1 = (torch.randn(I, H, device="cuda") * 0.02).to(bf); W2 = (torch.randn(H, I, device="cuda") * 0.02).to(bf)
    Wq = (torch.randn(3 * H, H, device="cuda") * 0.02).to(bf)
    b1 = torch.randn(I, device="cuda"); b2 = torch.randn(H, device="cuda"); bq = torch.randn(3 * H, device="cuda")
    u = torch.empty(M, I, device="cuda", dtype=bf); h = torch.empty(M, I, device="cuda", dtype=bf)
    qkv = torch.empty(M, 3 * H, device="cuda", dtype=bf)
    resid = torch.randn(M, H, device="cuda"); y = torch.empty(M, H, device="cuda"); yb = torch.empty(M, H, device="cuda", dtype=bf)
    xh = torch.empty(M, H, device="cuda", dtype=bf); rs = torch.empty(M, device="cuda")
    gamma = torch.ones(H, device="cuda"); beta = torch.zeros(H, device="cuda")
    part = torch.zeros((M + 127) // 128, 2, H, device="cuda")
    e0 = _lib.QstLnEpi(); e0.gamma, e0.beta, e0.eps, e0.xhat, e0.rstd = gamma.data_ptr(), beta.data_ptr(), 1e-12, xh.data_ptr(), rs.data_ptr()
    e1 = _lib.QstLnEpi(); e1.gamma, e1.xhat, e1.rstd, e1.partials = gamma.data_ptr(), xh.data_ptr(), rs.data_ptr(), part.data_ptr()
    cases = [
        ("QKV (N 2304, K 768, 16-bit out)", lambda: lib.qst_gemm_nt(gargs(A=x, B=Wq, C=qkv, bias=bq, M=M, N=3 * H, K=H, lda=H, ldb=H, ldc=3 * H), 0, st)),
        ("FFN-1 + GELU (N 3072, K 768)", lambda: lib.qst_gemm_nt(gargs(A=x, B=W1, C=u, C2=h, bias=b1, M=M, N=I, K=H, lda=H, ldb=H, ldc=I), 2, st)),
        ("GELU' dgrad (N 3072, K 768)", lambda: lib.qst_gemm_nt(gargs(A=x, B=W1, C=h, aux=u, M=M, N=I, K=H, lda=H, ldb=H, ldc=I), 3, st)),
        ("FFN-2 + LayerNorm (K 3072)", lambda: lib.qst_gemm_nt_ln(gargs(A=xi, B=W2, C=y, C2=yb, bias=b2, resid=resid, M=M, N=H, K=I, lda=I, ldb=I, ldc=H, ldr=H), e0, 0, st)),
        ("out-proj + LayerNorm (K 768)", lambda: lib.qst_gemm_nt_ln(gargs(A=x, B=Wq, C=y, C2=yb, bias=b2, resid=resid, M=M, N=H, K=H, lda=H, ldb=H, ldc=H, ldr=H), e0, 0, st)),
        ("FFN-1 dgrad + LayerNorm' (K 3072)", lambda: lib.qst_gemm_nt_ln(gargs(A=xi, B=W2, C=y, C2=yb, resid=resid, M=M, N=H, K=I, lda=I, ldb=I, ldc=H, ldr=H), e1, 1, st)),
    ]
    lib.qst_gemm_nt_ln(gargs(A=x, B=Wq, C=y, C2=yb, bias=b2, resid=resid, M=M, N=H, K=H, lda=H, ldb=H, ldc=H, ldr=H), e0, 0, st)   # xhat, rstd
    print(f"M = {M}; us per launch, best of 3 x 10, by spread of the first round's start (cycles)")
    print(f"{'launch':36s}" + "".join(f"{s:>10d}" for s in spreads))
    for name, fn in cases:
        row = []
        best = {s: 1e9 for s in spreads}
        for _ in range(3):
            for s in spreads:
                lib.qst_gemm8_stagger(s)
                best[s] = min(best[s], timeit(lambda: _lib.check(fn()), reps=10))
        print(f"{name:36s}" + "".join(f"{best[s]:10.1f}" for s in spreads), flush=True)
    lib.qst_gemm8_stagger(0)


if __name__ == "__main__":
    main()
