#!/usr/bin/env python3
"""MXFP8 NT GEMM: the tiled kernel (qst_gemm_nt_f8, 32x32x64 MFMA, 128 x 192 tile, two workgroups per CU) against the 8-phase
form (qst_gemm_nt8_f8, 16x16x128 MFMA, 128 x 384 / 256 x 256 tile) -- results compared, alternating best-of timing.

    python tools/f8_8phase_bench.py [M]      (default 196608 = bert-base configs[4]: 128 quadruplets x 4 x 384 tokens)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402


def timeit(fns, rounds=4, iters=5):
    best = [1e9] * len(fns)
    for _ in range(rounds):
        for i, f in enumerate(fns):
            f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            best[i] = min(best[i], e0.elapsed_time(e1) / iters * 1e3)
    return best


def main():
    lib = _lib.load()
    st = _lib.current_stream_ptr()
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 196608
    dev, bf = "cuda", torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(3)
    for name, N, K, epi in [("QKV", 2304, 768, 0), ("out", 768, 768, 1), ("FFN1", 3072, 768, 2), ("FFN1mx", 3072, 768, 6), ("FFN2", 768, 3072, 1),
                            ("ragged", 1000, 896, 0)]:
        Mx = M if name != "ragged" else 3000
        A = torch.randn(Mx, K, device=dev, generator=g)
        W = torch.randn(N, K, device=dev, generator=g) * 0.02
        bias = torch.randn(N, device=dev, generator=g) * 0.1
        resid = torch.randn(Mx, N, device=dev, generator=g) if epi == 1 else None
        Aq = torch.empty(Mx, K, dtype=torch.uint8, device=dev); As = torch.zeros(K // 128 * Mx * 4, dtype=torch.uint8, device=dev)
        Wq = torch.empty(N, K, dtype=torch.uint8, device=dev); Ws = torch.zeros(K // 128 * N * 4, dtype=torch.uint8, device=dev)
        _lib.check(lib.qst_quant_mx(A.data_ptr(), 0, Mx, K, Aq.data_ptr(), As.data_ptr(), st))
        _lib.check(lib.qst_quant_mx(W.data_ptr(), 0, N, K, Wq.data_ptr(), Ws.data_ptr(), st))
        outs = []

        def mk(which):
            C = torch.zeros(Mx, N, device=dev, dtype=torch.float32 if epi == 1 else bf)
            C2 = torch.zeros(Mx, N, device=dev, dtype=bf)
            f = _lib.QstGemmArgs()
            f.A, f.B, f.aux, f.bscale, f.bias = Aq.data_ptr(), Wq.data_ptr(), As.data_ptr(), Ws.data_ptr(), bias.data_ptr()
            f.C, f.C2 = C.data_ptr(), C2.data_ptr()
            if epi == 6:                         # the training FFN-1: + the bf16-rounded h as MXFP8
                C3 = torch.zeros(Mx, N, device=dev, dtype=torch.uint8)
                C4 = torch.zeros(N // 128 * Mx * 4, device=dev, dtype=torch.uint8)
                f.C3, f.C4 = C3.data_ptr(), C4.data_ptr()
                f._keep = (C3, C4)
                f.splits = 0x80 if which == 0 else 0
            f.resid = resid.data_ptr() if resid is not None else None
            f.M, f.N, f.K, f.lda, f.ldb, f.ldc, f.ldr = Mx, N, K, K, K, N, N
            outs.append((C, C2))
            if which == 0:
                return lambda: _lib.check(lib.qst_gemm_nt_f8(f, epi, st))
            return lambda: _lib.check(lib.qst_gemm_nt8_f8(f, epi, which - 1, st))
        fns = [mk(0), mk(1), mk(2)]
        for f in fns:
            f()
        torch.cuda.synchronize()
        ref = outs[0][0].float()
        d = [((o[0].float() - ref).abs().max().item(), ((o[0].float() - ref).norm() / ref.norm()).item()) for o in outs[1:]]
        t = timeit(fns)
        fl = 2.0 * Mx * N * K
        print(f"{name:6s} M={Mx} N={N:5d} K={K:5d} epi {epi}: tiled {t[0]:7.1f} us ({fl / t[0] / 1e6:6.0f} TF)   8-phase 128x384 {t[1]:7.1f} us "
              f"({fl / t[1] / 1e6:6.0f} TF)   256x256 {t[2]:7.1f} us ({fl / t[2] / 1e6:6.0f} TF)   max|d| {d[0][0]:.3g} / {d[1][0]:.3g}, "
              f"rel L2 {d[0][1]:.2e} / {d[1][1]:.2e}")


if __name__ == "__main__":
    main()
