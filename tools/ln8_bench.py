#!/usr/bin/env python3
"""GEMM + LayerNorm at H = 768: the unfused pair (qst_gemm_nt with the residual epilogue, then qst_ln_fwd / qst_ln_bwd)
against qst_gemm_nt_ln's several-tiles-per-row form (gemm8.hip: the three workgroups of a 256-row panel exchange the row
statistics inside the launch). usage: ln8_bench.py [M ...]   (default: 49152 196608 = configs[2] / configs[4] token rows)"""
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
N = 768


def gargs(**kw):
    g = _lib.QstGemmArgs()
    for k, v in kw.items():
        setattr(g, k, v.data_ptr() if torch.is_tensor(v) else v)
    return g


def main():
    Ms = [int(a) for a in sys.argv[1:]] or [49152, 196608]
    for M in Ms:
        gamma = 1 + 0.1 * torch.randn(N, device="cuda"); beta = 0.1 * torch.randn(N, device="cuda"); bias = torch.randn(N, device="cuda")
        resid = torch.randn(M, N, device="cuda")
        s = torch.empty(M, N, device="cuda"); y = torch.empty(M, N, device="cuda")
        yb = torch.empty(M, N, device="cuda", dtype=bf); xh = torch.empty(M, N, device="cuda", dtype=bf); rs = torch.empty(M, device="cuda")
        y2 = torch.empty(M, N, device="cuda"); yb2 = torch.empty(M, N, device="cuda", dtype=bf)
        xh2 = torch.empty(M, N, device="cuda", dtype=bf); rs2 = torch.empty(M, device="cuda")
        dgam = torch.zeros(N, device="cuda"); dbet = torch.zeros(N, device="cuda")
        scratch = torch.empty(lib.qst_ln_bwd_scratch_bytes(M, N) // 4, device="cuda")
        part = torch.zeros((M + 127) // 128, 2, N, device="cuda")
        for K in (768, 2304, 3072):
            A = torch.randn(M, K, device="cuda").to(bf); B = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
            e = _lib.QstLnEpi()
            e.gamma, e.beta, e.eps, e.xhat, e.rstd = gamma.data_ptr(), beta.data_ptr(), 1e-12, xh2.data_ptr(), rs2.data_ptr()

            def unf_fwd():
                _lib.check(lib.qst_gemm_nt(gargs(A=A, B=B, C=s, bias=bias, resid=resid, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, ldr=N), 1, st))
                _lib.check(lib.qst_ln_fwd(s.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-12, M, N, y.data_ptr(), yb.data_ptr(),
                                          xh.data_ptr(), rs.data_ptr(), st))

            def fus_fwd():
                _lib.check(lib.qst_gemm_nt_ln(gargs(A=A, B=B, C=y2, C2=yb2, bias=bias, resid=resid, M=M, N=N, K=K, lda=K, ldb=K, ldc=N,
                                                    ldr=N), e, 0, st))

            def gemm_only():
                _lib.check(lib.qst_gemm_nt(gargs(A=A, B=B, C=s, bias=bias, resid=resid, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, ldr=N), 1, st))
            unf_fwd(); fus_fwd(); torch.cuda.synchronize()
            d = (y2 - y).abs().max().item()
            e1 = _lib.QstLnEpi()
            e1.gamma, e1.xhat, e1.rstd, e1.partials = gamma.data_ptr(), xh.data_ptr(), rs.data_ptr(), part.data_ptr()

            def unf_bwd():
                _lib.check(lib.qst_gemm_nt(gargs(A=A, B=B, C=s, resid=resid, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, ldr=N), 1, st))
                _lib.check(lib.qst_ln_bwd(s.data_ptr(), xh.data_ptr(), rs.data_ptr(), gamma.data_ptr(), M, N, y.data_ptr(), yb.data_ptr(),
                                          dgam.data_ptr(), dbet.data_ptr(), scratch.data_ptr(), st))

            def fus_bwd():
                _lib.check(lib.qst_gemm_nt_ln(gargs(A=A, B=B, C=y2, C2=yb2, resid=resid, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, ldr=N), e1, 1, st))
            unf_bwd(); fus_bwd(); torch.cuda.synchronize()
            db = (y2 - y).abs().max().item() / max(1e-30, y.abs().max().item())
            best = {}
            for _ in range(3):
                for name, fn in (("gemm", gemm_only), ("unf_fwd", unf_fwd), ("fus_fwd", fus_fwd), ("unf_bwd", unf_bwd), ("fus_bwd", fus_bwd)):
                    best[name] = min(best.get(name, 1e9), timeit(fn, reps=10))
            print(f"M={M:7d} K={K:5d}: GEMM alone {best['gemm']:7.1f} us | forward: pair {best['unf_fwd']:7.1f} fused {best['fus_fwd']:7.1f} "
                  f"({best['fus_fwd'] / best['unf_fwd'] - 1:+.1%}) | backward: pair {best['unf_bwd']:7.1f} fused {best['fus_bwd']:7.1f} "
                  f"({best['fus_bwd'] / best['unf_bwd'] - 1:+.1%}) | max|dy| fwd {d:.1e}, bwd rel {db:.1e}", flush=True)
    print("exchange timeouts:", lib.qst_gemm_nt8_ln_timeouts())


if __name__ == "__main__":
    main()
