#!/usr/bin/env python3
"""Where the roles of the persistent ping-pong NT GEMM (csrc/gemm_pp.hip) spend their cycles: runs the diagnostic build
tools/libqst_stamp.so (hipcc ... -DQST_PP_STAMP, see csrc/gemm_pp.hip) on the step's shapes and prints, per role, the median
over workgroups of the summed shader cycles.  usage: pp_stamp.py [M] [H]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 384
    I = 4 * H
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("PP_STAMP_LIB", "libqst_stamp.so")))
    res, args = _lib.SIGNATURES["qst_gemm_nt"]
    lib.qst_gemm_nt.restype, lib.qst_gemm_nt.argtypes = res, args
    st = _lib.current_stream_ptr()
    bf = torch.bfloat16
    cases = [("QKV fwd  epi0", 3 * H, H, 0), ("FFN1 fwd epi2", I, H, 2), ("FFN2 dgrad epi3", I, H, 3), ("out dgrad epi0", H, H, 0),
             ("FFN2 fwd epi1", H, I, 1)]
    for name, N, K, epi in cases:
        A = torch.randn(M, K, device="cuda").to(bf)
        B = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
        bias = torch.randn(N, device="cuda")
        resid = torch.randn(M, N, device="cuda")
        aux = torch.randn(M, N, device="cuda").to(bf)
        Cm = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == 1 else bf)
        C2 = torch.empty(M, N, device="cuda", dtype=bf)
        stamps = torch.zeros(256 * 12, dtype=torch.int64, device="cuda")
        g = _lib.QstGemmArgs()
        g.A, g.B, g.C, g.C2, g.aux, g.bias, g.resid = (A.data_ptr(), B.data_ptr(), Cm.data_ptr(), C2.data_ptr(), aux.data_ptr(),
                                                        bias.data_ptr(), resid.data_ptr())
        g.colsum = stamps.data_ptr()
        g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldr, g.splits = M, N, K, K, K, N, N, 5
        for _ in range(5):
            _lib.check(lib.qst_gemm_nt(g, epi, st))
        torch.cuda.synchronize()
        s = stamps.view(256, 3, 4).double().cpu()
        med = s.median(dim=0).values
        nk = K // 64
        tiles = ((M + 127) // 128) * ((N + 191) // 192)
        steps = (tiles / 256) * nk
        print(f"{name}  N={N} K={K}: ~{steps:.0f} K stages per workgroup; kernel {med[0][3]:.0f} cycles = {med[0][3] / max(steps, 1):.0f} per stage")
        print(f"   loader : DMA issue {med[0][0]:9.0f}   vmcnt wait {med[0][1]:9.0f}   barrier wait {med[0][2]:9.0f}")
        print(f"   group 0: K stages  {med[1][0]:9.0f}   epilogue   {med[1][1]:9.0f}   barrier wait {med[1][2]:9.0f}")
        print(f"   group 1: K stages  {med[2][0]:9.0f}   epilogue   {med[2][1]:9.0f}   barrier wait {med[2][2]:9.0f}")


if __name__ == "__main__":
    main()
