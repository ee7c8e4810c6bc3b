#!/usr/bin/env python3
"""Does the row stride of a GEMM operand matter (L2 / HBM channel aliasing)? The K = 1536 launches of the MiniLM step with the
operands' rows K elements apart (3,072 bytes = 12 x 256) against K + pad elements apart.   python tools/stride_probe.py [pad]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402


def timeit(fns, rounds=5, iters=10):
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
    pad = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    lib = _lib.load()
    st = _lib.current_stream_ptr()
    dev, bf = "cuda", torch.bfloat16
    M, H = 32768, 384
    for K in (1536, 384, 1152):
        outs = []

        def mk(pa, pb, fused):
            A = torch.randn(M, K + pa, device=dev).to(bf)
            B = (torch.randn(H, K + pb, device=dev) * 0.02).to(bf)
            bias = torch.zeros(H, device=dev); gamma = torch.ones(H, device=dev); beta = torch.zeros(H, device=dev)
            resid = torch.randn(M, H, device=dev)
            y = torch.empty(M, H, device=dev); yb = torch.empty(M, H, device=dev, dtype=bf); xh = torch.empty(M, H, device=dev, dtype=bf)
            rs = torch.empty(M, device=dev)
            g = _lib.QstGemmArgs()
            g.A, g.B, g.C, g.C2, g.bias, g.resid = A.data_ptr(), B.data_ptr(), y.data_ptr(), yb.data_ptr(), bias.data_ptr(), resid.data_ptr()
            g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldr = M, H, K, K + pa, K + pb, H, H
            e = _lib.QstLnEpi()
            e.gamma, e.beta, e.eps, e.xhat, e.rstd = gamma.data_ptr(), beta.data_ptr(), 1e-12, xh.data_ptr(), rs.data_ptr()
            g._keep = (A, B, bias, gamma, beta, resid, y, yb, xh, rs)
            outs.append(g)
            if fused:
                return lambda: _lib.check(lib.qst_gemm_nt_ln(g, e, 0, st))
            return lambda: _lib.check(lib.qst_gemm_nt(g, 1, st))
        for fused in (True, False):
            t = timeit([mk(0, 0, fused), mk(pad, 0, fused), mk(0, pad, fused), mk(pad, pad, fused)])
            print(f"K = {K:5d} {'GEMM + LayerNorm' if fused else 'GEMM fp32 + residual'}: rows K apart {t[0]:6.1f} us   A rows K + {pad} apart "
                  f"{t[1]:6.1f}   B rows K + {pad} apart {t[2]:6.1f}   both {t[3]:6.1f}")


if __name__ == "__main__":
    main()
