#!/usr/bin/env python3
"""Does the Infinity Cache (256 MB, memory side) explain why the fused FFN-2 + LayerNorm launch takes 90 us inside the
training step and 57 us when timed back to back on the same buffers? One launch each, timed with events, after:
  warm        the same launch just before (operands of the previous launch still in the cache)
  cold        600 MB of unrelated writes in between
  cold+read   cold, then a sequential read of the activation operand A (a reduction kernel), then the GEMM
  cold+write  cold, then A is (re)written by a copy kernel -- what the producing GEMM does inside the step
Prints the GEMM's time alone and, for the last two, the time including the extra kernel.

Measured (round 3): K = 1536: warm 58, cold 97, cold+read 72, cold+write 93 us; K = 384: 34 / 56 / 50 / 55. Inside the step the
launch takes 90 / 42 us: it runs "cold" -- what the previous kernel wrote is not served from the cache. Two remedies tried in
nt_mainloop and removed: an L2 touch-ahead of A three stages early (cold 111 us, warm 62: worse) and a tile-blocked A layout
[M/128][K/64][128][64] that makes every stage one contiguous 16 KB read (cold 94 vs 98: the access pattern is not it). A padded row pitch (K + 8 ...
K + 72 elements) changes nothing (cold 91-99), nor do touches issued by a ninth wave that never waits for them."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402

lib = _lib.load()
st = _lib.current_stream_ptr()
M, H = 32768, 384
dev, bf = "cuda", torch.bfloat16
junk = torch.empty(600 * 1024 * 1024, dtype=torch.uint8, device=dev)


def ev():
    return torch.cuda.Event(enable_timing=True)


for K in (1536, 384):
    A = torch.randn(M, K, device=dev).to(bf); A2 = A.clone(); B = (torch.randn(H, K, device=dev) * 0.02).to(bf)
    bias = torch.zeros(H, device=dev); gamma = torch.ones(H, device=dev); beta = torch.zeros(H, device=dev)
    resid = torch.randn(M, H, device=dev)
    y = torch.empty(M, H, device=dev); yb = torch.empty(M, H, device=dev, dtype=bf); xh = torch.empty(M, H, device=dev, dtype=bf)
    rs = torch.empty(M, device=dev)
    g = _lib.QstGemmArgs()
    g.A, g.B, g.C, g.C2, g.bias, g.resid = A.data_ptr(), B.data_ptr(), y.data_ptr(), yb.data_ptr(), bias.data_ptr(), resid.data_ptr()
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldr = M, H, K, K, K, H, H
    e = _lib.QstLnEpi()
    e.gamma, e.beta, e.eps, e.xhat, e.rstd = gamma.data_ptr(), beta.data_ptr(), 1e-12, xh.data_ptr(), rs.data_ptr()

    def gemm():
        _lib.check(lib.qst_gemm_nt_ln(g, e, 0, st))

    def run(prep):
        ts, tt = [], []
        for _ in range(7):
            e0, e1, e2 = ev(), ev(), ev()
            if prep != "warm":
                junk.fill_(1)
            else:
                gemm()
            e0.record()
            if prep == "cold+read":
                A.view(torch.int32).sum()
            elif prep == "cold+write":
                A.copy_(A2)
            e1.record()
            gemm()
            e2.record()
            torch.cuda.synchronize()
            ts.append(e1.elapsed_time(e2) * 1e3); tt.append(e0.elapsed_time(e2) * 1e3)
        return sorted(ts)[len(ts) // 2], sorted(tt)[len(tt) // 2]
    out = []
    for prep in ("warm", "cold", "cold+read", "cold+write"):
        a, b = run(prep)
        out.append(f"{prep}: {a:.1f} us" + (f" ({b:.1f} with the extra kernel)" if "+" in prep else ""))
    print(f"gemm_nt_ln forward M={M} K={K}: " + "   ".join(out))
