#!/usr/bin/env python3
"""Same-process A/B of the NT GEMM forms (QstGemmArgs.splits: 1 = 128-row tiles, 4 = tall, 5 = persistent ping-pong,
0 = automatic) on the step's shapes.  usage: pp_bench.py [M] [H]"""
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
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 384
    I = 4 * H
    lib = _lib.load()
    st = _lib.current_stream_ptr()
    bf = torch.bfloat16
    cases = [("QKV fwd  epi0", 3 * H, H, 0), ("FFN1 fwd epi2", I, H, 2), ("FFN2 dgrad epi3", I, H, 3), ("out dgrad epi0", H, H, 0),
             ("out fwd  epi1", H, H, 1), ("FFN2 fwd epi1", H, I, 1), ("FFN1 dgrad epi1", H, I, 1), ("QKV dgrad epi1", H, 3 * H, 1)]
    forms = [1, 4, 5, 0]
    tot = {f: 0.0 for f in forms}
    print(f"M={M} H={H}: us per launch by form {forms}")
    for name, N, K, epi in cases:
        A = torch.randn(M, K, device="cuda").to(bf)
        B = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
        bias = torch.randn(N, device="cuda")
        resid = torch.randn(M, N, device="cuda")
        aux = torch.randn(M, N, device="cuda").to(bf)
        outs = {}
        best = {f: 1e9 for f in forms}
        g = _lib.QstGemmArgs()
        Cm = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == 1 else bf)
        C2 = torch.empty(M, N, device="cuda", dtype=bf)
        g.A, g.B, g.C, g.C2, g.aux, g.bias, g.resid = (A.data_ptr(), B.data_ptr(), Cm.data_ptr(), C2.data_ptr(), aux.data_ptr(),
                                                        bias.data_ptr(), resid.data_ptr())
        g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldr = M, N, K, K, K, N, N
        for f in forms:
            g.splits = f
            Cm.zero_(); C2.zero_()
            _lib.check(lib.qst_gemm_nt(g, epi, st))
            torch.cuda.synchronize()
            outs[f] = (Cm.clone(), C2.clone())
        same = all(torch.equal(outs[1][0], outs[f][0]) and (epi != 2 or torch.equal(outs[1][1], outs[f][1])) for f in forms)
        for _ in range(3):
            for f in forms:
                g.splits = f
                best[f] = min(best[f], timeit(lambda: _lib.check(lib.qst_gemm_nt(g, epi, st))))
        for f in forms:
            tot[f] += best[f]
        fl = 2.0 * M * N * K
        print(f"{name:18s} N={N:5d} K={K:5d}  " + "  ".join(f"f{f}: {best[f]:7.1f} us {fl / best[f] / 1e6:6.0f} TF" for f in forms) +
              f"   bit-identical: {same}")
    print("sum                " + "  ".join(f"f{f}: {tot[f]:7.1f}" for f in forms))


if __name__ == "__main__":
    main()
