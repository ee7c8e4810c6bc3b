#!/usr/bin/env python3
"""A/B two builds of libqst.so on the fused GEMM + LayerNorm kernel (forward and backward epilogues) at the step's shapes,
in ONE process, launches of the two builds interleaved. usage: ab_ln.py old.so [new.so]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402


def bind(path):
    lib = C.CDLL(path)
    res, args = _lib.SIGNATURES["qst_gemm_nt_ln"]
    lib.qst_gemm_nt_ln.restype, lib.qst_gemm_nt_ln.argtypes = res, args
    return lib


def main():
    libs = [bind(os.path.abspath(sys.argv[1])), bind(os.path.abspath(sys.argv[2]) if len(sys.argv) > 2 else _lib.LIB_PATH)]
    st = _lib.current_stream_ptr()
    M, N = 32768, 384
    dev, bf = "cuda", torch.bfloat16
    for mode, K in ((0, 384), (0, 1536), (1, 1536), (1, 1152)):
        A = torch.randn(M, K, device=dev).to(bf); B = (torch.randn(N, K, device=dev) * 0.02).to(bf)
        resid = torch.randn(M, N, device=dev); bias = torch.randn(N, device=dev)
        gamma = torch.ones(N, device=dev); beta = torch.zeros(N, device=dev)
        Cf = torch.empty(M, N, device=dev); C2 = torch.empty(M, N, device=dev, dtype=bf)
        xh = torch.randn(M, N, device=dev).to(bf); rs = torch.rand(M, device=dev) + 0.5
        part = torch.empty((M + 127) // 128, 2, N, device=dev)
        g = _lib.QstGemmArgs()
        g.A, g.B, g.C, g.C2, g.resid = A.data_ptr(), B.data_ptr(), Cf.data_ptr(), C2.data_ptr(), resid.data_ptr()
        g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldr = M, N, K, K, K, N, N
        e = _lib.QstLnEpi()
        e.gamma, e.eps, e.xhat, e.rstd = gamma.data_ptr(), 1e-12, xh.data_ptr(), rs.data_ptr()
        if mode == 0:
            g.bias, e.beta = bias.data_ptr(), beta.data_ptr()
        else:
            e.partials = part.data_ptr()
        outs, best = [], [1e9, 1e9]
        for rnd in range(5):
            for i, lib in enumerate(libs):
                for _ in range(3):
                    _lib.check(lib.qst_gemm_nt_ln(g, e, mode, st))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    _lib.check(lib.qst_gemm_nt_ln(g, e, mode, st))
                e1.record()
                torch.cuda.synchronize()
                best[i] = min(best[i], e0.elapsed_time(e1) / 20 * 1e3)
                if rnd == 0:
                    outs.append((Cf.clone(), C2.clone()))
        same = all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
        print(f"mode {mode} K={K:5d}: old {best[0]:7.1f} us   new {best[1]:7.1f} us   ({100 * (best[1] / best[0] - 1):+.1f}%)   outputs identical: {same}")


if __name__ == "__main__":
    main()
