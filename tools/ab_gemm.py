#!/usr/bin/env python3
"""A/B two builds of libqst.so on the step's NT GEMM shapes in ONE process (box-to-box variance is +-5%).
usage: ab_gemm.py old.so [new.so]   (new defaults to the in-tree library)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402


def bind(path):
    lib = C.CDLL(path)
    res, args = _lib.SIGNATURES["qst_gemm_nt"]
    lib.qst_gemm_nt.restype, lib.qst_gemm_nt.argtypes = res, args
    return lib


def timeit(fn, reps=30):
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
    libs = [bind(os.path.abspath(sys.argv[1])), bind(os.path.abspath(sys.argv[2]) if len(sys.argv) > 2 else _lib.LIB_PATH)]
    st = _lib.current_stream_ptr()
    M, H, I = 32768, 384, 1536
    bf = torch.bfloat16
    cases = [("QKV fwd  epi0", 3 * H, H, 0), ("out fwd  epi1", H, H, 1), ("FFN1 fwd epi2", I, H, 2),
             ("FFN2 fwd epi1", H, I, 1), ("FFN2 dgrad epi3", I, H, 3), ("FFN1 dgrad epi1", H, I, 1),
             ("out dgrad epi0", H, H, 0), ("QKV dgrad epi1", H, 3 * H, 1)]
    tot = [0.0, 0.0]
    for name, N, K, epi in cases:
        A = torch.randn(M, K, device="cuda").to(bf)
        B = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
        bias = torch.zeros(N, device="cuda")
        resid = torch.randn(M, N, device="cuda")
        aux = torch.randn(M, N, device="cuda").to(bf)
        Cm = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == 1 else bf)
        C2 = torch.empty(M, N, device="cuda", dtype=bf)
        g = _lib.QstGemmArgs()
        g.A, g.B, g.C, g.C2, g.aux, g.bias, g.resid = (A.data_ptr(), B.data_ptr(), Cm.data_ptr(), C2.data_ptr(),
                                                        aux.data_ptr(), bias.data_ptr(), resid.data_ptr())
        g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldr = M, N, K, K, K, N, N
        g.splits = 0
        best = [1e9, 1e9]
        for _ in range(3):
            for i, lib in enumerate(libs):
                best[i] = min(best[i], timeit(lambda: _lib.check(lib.qst_gemm_nt(g, epi, st))))
        tot[0] += best[0]; tot[1] += best[1]
        print(f"{name:18s} old {best[0]:7.1f} us   new {best[1]:7.1f} us   ({best[1] / best[0] - 1:+.1%})")
    print(f"sum                old {tot[0]:7.1f} us   new {tot[1]:7.1f} us")


if __name__ == "__main__":
    main()
