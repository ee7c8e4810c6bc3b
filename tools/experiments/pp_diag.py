#!/usr/bin/env python3
"""Time the ping-pong NT GEMM (form 5) of several builds of libqst.so on the bf16-output shapes of the step.
usage: pp_diag.py lib1.so [lib2.so ...]   (paths relative to tools/; "product" = the in-tree library)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402
from pp_bench import timeit  # noqa: E402


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    libs = []
    for name in sys.argv[1:]:
        lib = C.CDLL(_lib.LIB_PATH if name == "product" else os.path.join(here, name))
        res, args = _lib.SIGNATURES["qst_gemm_nt"]
        lib.qst_gemm_nt.restype, lib.qst_gemm_nt.argtypes = res, args
        libs.append((name, lib))
    M, H = 32768, 384
    I = 4 * H
    st = _lib.current_stream_ptr()
    bf = torch.bfloat16
    for cname, N, K, epi in [("QKV fwd  epi0", 3 * H, H, 0), ("FFN1 fwd epi2", I, H, 2), ("FFN2 dgrad epi3", I, H, 3), ("out dgrad epi0", H, H, 0)]:
        A = torch.randn(M, K, device="cuda").to(bf)
        B = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
        bias = torch.randn(N, device="cuda")
        aux = torch.randn(M, N, device="cuda").to(bf)
        Cm = torch.empty(M, N, device="cuda", dtype=bf)
        C2 = torch.empty(M, N, device="cuda", dtype=bf)
        g = _lib.QstGemmArgs()
        g.A, g.B, g.C, g.C2, g.aux, g.bias = A.data_ptr(), B.data_ptr(), Cm.data_ptr(), C2.data_ptr(), aux.data_ptr(), bias.data_ptr()
        g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldr, g.splits = M, N, K, K, K, N, N, 5
        out = []
        for name, lib in libs:
            best = min(timeit(lambda: _lib.check(lib.qst_gemm_nt(g, epi, st))) for _ in range(3))
            out.append(f"{name}: {best:6.1f} us")
        print(f"{cname:16s} " + "   ".join(out))


if __name__ == "__main__":
    main()
