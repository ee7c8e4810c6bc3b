#!/usr/bin/env python3
"""bf16 vs MXFP8 NT GEMM at the bert-base forward shapes (M = 196608 = 128 quadruplets x 4 x 384 tokens)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402
from gemm_bench import timeit  # noqa: E402


def main():
    lib = _lib.load()
    st = _lib.current_stream_ptr()
    M = int(os.environ.get("M", 196608))
    dev, bf = "cuda", torch.bfloat16
    for name, N, K, epi_b, epi_f in [("QKV", 2304, 768, 0, 0), ("out", 768, 768, 1, 1), ("FFN1", 3072, 768, 2, 5), ("FFN2", 768, 3072, 1, 1)]:
        A = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) * 0.02
        bias = torch.zeros(N, device=dev)
        resid = torch.randn(M, N, device=dev) if epi_b == 1 else None
        Ab, Wb = A.to(bf), W.to(bf)
        C = torch.empty(M, N, device=dev, dtype=torch.float32 if epi_b == 1 else bf)
        C2 = torch.empty(M, N, device=dev, dtype=bf)
        g = _lib.QstGemmArgs()
        g.A, g.B, g.C, g.C2, g.bias = Ab.data_ptr(), Wb.data_ptr(), C.data_ptr(), C2.data_ptr(), bias.data_ptr()
        g.resid = resid.data_ptr() if resid is not None else None
        g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldr = M, N, K, K, K, N, N
        t_b = timeit(lambda: _lib.check(lib.qst_gemm_nt(g, epi_b, st)), reps=5)
        Aq = torch.empty(M, K, dtype=torch.uint8, device=dev); As = torch.zeros(K // 128 * M * 4, dtype=torch.uint8, device=dev)
        Wq = torch.empty(N, K, dtype=torch.uint8, device=dev); Ws = torch.zeros(K // 128 * N * 4, dtype=torch.uint8, device=dev)
        _lib.check(lib.qst_quant_mx(A.data_ptr(), 0, M, K, Aq.data_ptr(), As.data_ptr(), st))
        _lib.check(lib.qst_quant_mx(W.data_ptr(), 0, N, K, Wq.data_ptr(), Ws.data_ptr(), st))
        Hq = torch.empty(M, N, dtype=torch.uint8, device=dev); Hs = torch.zeros(max(1, N // 128) * M * 4, dtype=torch.uint8, device=dev)
        f = _lib.QstGemmArgs()
        f.A, f.B, f.aux, f.bscale, f.bias = Aq.data_ptr(), Wq.data_ptr(), As.data_ptr(), Ws.data_ptr(), bias.data_ptr()
        f.C, f.C2 = (Hq.data_ptr(), Hs.data_ptr()) if epi_f == 5 else (C.data_ptr(), None)
        f.resid = g.resid
        f.M, f.N, f.K, f.lda, f.ldb, f.ldc, f.ldr = M, N, K, K, K, N, N
        res = []
        for sp in (0, 1):
            f.splits = sp
            res.append(timeit(lambda: _lib.check(lib.qst_gemm_nt_f8(f, epi_f, st)), reps=5))
        fl = 2.0 * M * N * K
        print(f"{name:5s} N={N:5d} K={K:5d}: bf16 {t_b:7.1f} us ({fl / t_b / 1e6:6.1f} TF)   fp8 {res[0]:7.1f} us ({fl / res[0] / 1e6:6.1f} TF)"
              f"   fp8 without scale loads {res[1]:7.1f} us")


if __name__ == "__main__":
    main()
