#!/usr/bin/env python3
"""The tall 256 x 192 tile (four waves of 128 x 96; QstGemmArgs.splits = 4) against the 128 x 192 form (splits = 1) of
qst_gemm_nt on the K >= 768 shapes, for the bf16 (epi 0) and the fp32 + residual (epi 1) epilogue; outputs compared.

    python tools/big_tile_bench.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402


def timeit(fn, reps=10):
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
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 196608
    lib = _lib.load()
    st = _lib.current_stream_ptr()
    bf = torch.bfloat16
    for epi in (1, 0):
        for N, K in [(768, 3072), (768, 2304), (768, 768), (3072, 768)]:
            A = torch.randn(M, K, device="cuda").to(bf)
            B = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
            bias = torch.randn(N, device="cuda")
            resid = torch.randn(M, N, device="cuda")
            outs, res, gs = {}, {}, {}
            for name, force in (("128x192", 1), ("tall", 4), ("auto", 0)):
                C = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi == 1 else bf)
                g = _lib.QstGemmArgs()
                g.A, g.B, g.C, g.bias, g.resid = A.data_ptr(), B.data_ptr(), C.data_ptr(), bias.data_ptr(), resid.data_ptr()
                g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldr = M, N, K, K, K, N, N
                g.splits = force
                gs[name], outs[name] = g, C
            for _ in range(4):                         # alternating, best of four: the first launches of a shape run slower
                for name, g in gs.items():
                    res[name] = min(res.get(name, 1e9), timeit(lambda: _lib.check(lib.qst_gemm_nt(g, epi, st)), reps=5))
            d = (outs["tall"].float() - outs["128x192"].float()).abs().max().item()
            fl = 2.0 * M * N * K
            print(f"epi {epi} M={M} N={N} K={K}: " + ", ".join(f"{k} {v:.1f} us ({fl / v / 1e6:.0f} TF/s)" for k, v in res.items())
                  + f" | max |tall - 128x192| = {d:.3g}", flush=True)
            del A, B, outs, resid


if __name__ == "__main__":
    main()
