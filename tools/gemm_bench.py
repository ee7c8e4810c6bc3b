#!/usr/bin/env python3
"""Micro-benchmark of the step's GEMM launches at BASELINE configs[1] shapes (M = 4*64*128 = 32768)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    lib = _lib.load()
    st = _lib.current_stream_ptr()
    dev = "cuda"
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 384
    I = 4 * H
    bf = torch.bfloat16
    tot = 0.0
    print(f"{'launch':34s} {'us':>8s} {'TFLOP/s':>8s}")
    cases = [("QKV fwd  epi0", 3 * H, H, 0, 1), ("out fwd  epi1", H, H, 1, 1), ("FFN1 fwd epi2", I, H, 2, 1),
             ("FFN2 fwd epi1", H, I, 1, 1), ("FFN2 dgrad epi3", I, H, 3, 1), ("FFN1 dgrad epi1", H, I, 1, 1),
             ("out dgrad epi0", H, H, 0, 1), ("QKV dgrad epi1", H, 3 * H, 1, 1)]
    for name, N, K, epi, _ in cases:
        A = torch.randn(M, K, device=dev).to(bf)
        B = (torch.randn(N, K, device=dev) * 0.02).to(bf)
        bias = torch.zeros(N, device=dev)
        resid = torch.randn(M, N, device=dev)
        aux = torch.randn(M, N, device=dev).to(bf)
        C = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == 1 else bf)
        C2 = torch.empty(M, N, device=dev, dtype=bf)
        g = _lib.QstGemmArgs()
        g.A, g.B, g.C, g.C2, g.aux, g.bias, g.resid = A.data_ptr(), B.data_ptr(), C.data_ptr(), C2.data_ptr(), aux.data_ptr(), bias.data_ptr(), resid.data_ptr()
        g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldr = M, N, K, K, K, N, N
        res = []
        for force in (1, 2, 3, 4):
            g.splits = force
            res.append(timeit(lambda: _lib.check(lib.qst_gemm_nt(g, epi, st))))
        g.splits = 0
        us = timeit(lambda: _lib.check(lib.qst_gemm_nt(g, epi, st)))
        tot += us
        print(f"nt {name:31s} {us:8.1f} {2.0 * M * N * K / us / 1e6:8.1f}   (128-row tile {res[0]:.1f} us, 256-row tile {res[1]:.1f} us, 128x384 tile {res[2]:.1f} us, tall 256x192 {res[3]:.1f} us)")
    # fused GEMM + LayerNorm (N = 384 full-row tiles) against the unfused pair
    if lib.qst_gemm_nt_ln_supported(H):
        for name, K, mode in [("out+LN1 fwd", H, 0), ("FFN2+LN2 fwd", I, 0), ("FFN1 dgrad+LN1 bwd", I, 1), ("QKV dgrad+LN2 bwd", 3 * H, 1)]:
            A = torch.randn(M, K, device=dev).to(bf)
            B = (torch.randn(H, K, device=dev) * 0.02).to(bf)
            bias = torch.zeros(H, device=dev); gamma = torch.ones(H, device=dev); beta = torch.zeros(H, device=dev)
            resid = torch.randn(M, H, device=dev)
            s_ = torch.empty(M, H, device=dev); y = torch.empty(M, H, device=dev)
            yb = torch.empty(M, H, device=dev, dtype=bf); xh = torch.randn(M, H, device=dev).to(bf)
            rs = torch.rand(M, device=dev) + 0.5
            part = torch.empty((M + 127) // 128, 2, H, device=dev)
            scratch = torch.empty(lib.qst_ln_bwd_scratch_bytes(M, H) // 4, device=dev)
            g = _lib.QstGemmArgs()
            g.A, g.B, g.C, g.C2, g.bias, g.resid = A.data_ptr(), B.data_ptr(), y.data_ptr(), yb.data_ptr(), bias.data_ptr(), resid.data_ptr()
            g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldr = M, H, K, K, K, H, H
            e = _lib.QstLnEpi()
            e.gamma, e.beta, e.eps, e.xhat, e.rstd, e.partials = gamma.data_ptr(), beta.data_ptr(), 1e-12, xh.data_ptr(), rs.data_ptr(), part.data_ptr()
            g0 = _lib.QstGemmArgs()
            g0.A, g0.B, g0.C, g0.bias, g0.resid = A.data_ptr(), B.data_ptr(), s_.data_ptr(), bias.data_ptr(), resid.data_ptr()
            g0.M, g0.N, g0.K, g0.lda, g0.ldb, g0.ldc, g0.ldr = M, H, K, K, K, H, H

            def pair():
                _lib.check(lib.qst_gemm_nt(g0, 1, st))
                if mode == 0:
                    _lib.check(lib.qst_ln_fwd(s_.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-12, M, H, y.data_ptr(),
                                              yb.data_ptr(), xh.data_ptr(), rs.data_ptr(), st))
                else:
                    _lib.check(lib.qst_ln_bwd(s_.data_ptr(), xh.data_ptr(), rs.data_ptr(), gamma.data_ptr(), M, H, y.data_ptr(),
                                              yb.data_ptr(), None, None, scratch.data_ptr(), st))
            t_pair = timeit(pair)
            t_fused = timeit(lambda: _lib.check(lib.qst_gemm_nt_ln(g, e, mode, st)))
            print(f"ln {name:31s} fused {t_fused:7.1f} us   unfused pair {t_pair:7.1f} us")
    # the feed-forward block as one kernel (csrc/ffn.hip) against the two launches it replaces
    if lib.qst_ffn_chain_supported(H, I):
        A = torch.randn(M, H, device=dev).to(bf)
        W1 = (torch.randn(I, H, device=dev) * 0.02).to(bf); W2 = (torch.randn(H, I, device=dev) * 0.02).to(bf)
        b1 = torch.zeros(I, device=dev); b2 = torch.zeros(H, device=dev); gamma = torch.ones(H, device=dev); beta = torch.zeros(H, device=dev)
        resid = torch.randn(M, H, device=dev)
        gp = torch.empty(M, I, device=dev, dtype=bf); hh = torch.empty(M, I, device=dev, dtype=bf); du = torch.empty(M, I, device=dev, dtype=bf)
        y = torch.empty(M, H, device=dev); yb = torch.empty(M, H, device=dev, dtype=bf); xh = torch.empty(M, H, device=dev, dtype=bf)
        rs = torch.rand(M, device=dev) + 0.5
        part = torch.empty((M + 127) // 128, 2, H, device=dev)
        e = _lib.QstLnEpi()
        e.gamma, e.beta, e.eps, e.xhat, e.rstd, e.partials = gamma.data_ptr(), beta.data_ptr(), 1e-12, xh.data_ptr(), rs.data_ptr(), part.data_ptr()
        f = _lib.QstFfnArgs()
        f.A, f.B1, f.B2, f.bias1, f.bias2, f.resid = A.data_ptr(), W1.data_ptr(), W2.data_ptr(), b1.data_ptr(), b2.data_ptr(), resid.data_ptr()
        f.C, f.C2, f.M, f.H, f.I = y.data_ptr(), yb.data_ptr(), M, H, I
        f.save_gp, f.save_h = gp.data_ptr(), hh.data_ptr()
        t_tr = timeit(lambda: _lib.check(lib.qst_ffn_chain(f, e, 0, st)))
        f.save_gp, f.save_h = None, None
        t_inf = timeit(lambda: _lib.check(lib.qst_ffn_chain(f, e, 0, st)))
        f.aux, f.save_h, f.bias1, f.bias2 = gp.data_ptr(), du.data_ptr(), None, None
        t_bw = timeit(lambda: _lib.check(lib.qst_ffn_chain(f, e, 1, st)))
        fl = 4.0 * M * H * I
        print(f"ffn chain fwd (training, saves gelu' + h)  {t_tr:8.1f} us {fl / t_tr / 1e6:8.1f} TF   inference {t_inf:8.1f} us {fl / t_inf / 1e6:8.1f} TF   "
              f"bwd (du + LN1') {t_bw:8.1f} us {fl / t_bw / 1e6:8.1f} TF")
    for name, N, K in [("dW2 [H,I]", H, I), ("dW1 [I,H]", I, H), ("dWo [H,H]", H, H), ("dWqkv [3H,H]", 3 * H, H)]:
        A = torch.randn(M, N, device=dev).to(bf)
        B = torch.randn(M, K, device=dev).to(bf)
        C = torch.zeros(N, K, device=dev)
        cs = torch.zeros(N, device=dev)
        g = _lib.QstGemmArgs()
        g.A, g.B, g.C, g.colsum = A.data_ptr(), B.data_ptr(), C.data_ptr(), cs.data_ptr()
        g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.splits = M, N, K, N, K, K, 0
        us = timeit(lambda: _lib.check(lib.qst_gemm_tn(g, st)))
        tot += us
        print(f"tn {name:31s} {us:8.1f} {2.0 * M * N * K / us / 1e6:8.1f}")
    print(f"sum per layer {tot:.1f} us -> x6 layers = {tot * 6 / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
