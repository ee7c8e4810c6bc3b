#!/usr/bin/env python3
"""Timing experiment on gemm_nt_ln_kernel<0> (FFN2 + LayerNorm-2 shape): which epilogue stores cost what.
QstGemmArgs.splits bits (diagnostic): 1 = no bf16 y store, 2 = no xhat store, 4 = no fp32 y store."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402
from gemm_bench import timeit  # noqa: E402

lib = _lib.load()
st = _lib.current_stream_ptr()
M, H = 32768, 384
dev, bf = "cuda", torch.bfloat16
for K in (1536, 384):
    A = torch.randn(M, K, device=dev).to(bf); B = (torch.randn(H, K, device=dev) * 0.02).to(bf)
    bias = torch.zeros(H, device=dev); gamma = torch.ones(H, device=dev); beta = torch.zeros(H, device=dev)
    resid = torch.randn(M, H, device=dev)
    y = torch.empty(M, H, device=dev); yb = torch.empty(M, H, device=dev, dtype=bf); xh = torch.empty(M, H, device=dev, dtype=bf)
    rs = torch.empty(M, device=dev)
    g = _lib.QstGemmArgs()
    g.A, g.B, g.C, g.C2, g.bias, g.resid = A.data_ptr(), B.data_ptr(), y.data_ptr(), yb.data_ptr(), bias.data_ptr(), resid.data_ptr()
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldr = M, H, K, K, K, H, H
    e = _lib.QstLnEpi()
    e.gamma, e.beta, e.eps, e.xhat, e.rstd = gamma.data_ptr(), beta.data_ptr(), 1e-12, xh.data_ptr(), rs.data_ptr()
    out = []
    for bits in (0, 1, 2, 3, 4, 7):
        g.splits = bits
        out.append(f"bits={bits}: {min(timeit(lambda: _lib.check(lib.qst_gemm_nt_ln(g, e, 0, st))) for _ in range(3)):6.1f} us")
    g.resid = None
    g.splits = 0
    out.append(f"no resid read: {min(timeit(lambda: _lib.check(lib.qst_gemm_nt_ln(g, e, 0, st))) for _ in range(3)):6.1f} us")
    g.splits = 7
    out.append(f"no resid, no stores: {min(timeit(lambda: _lib.check(lib.qst_gemm_nt_ln(g, e, 0, st))) for _ in range(3)):6.1f} us")
    print(f"K={K}: " + "   ".join(out))

# phase stamps (s_memtime, wave 0 of every workgroup): bit 3 of splits, stamps written to the `partials` buffer
import numpy as np
for K in (1536, 384):
    A = torch.randn(M, K, device=dev).to(bf); B = (torch.randn(H, K, device=dev) * 0.02).to(bf)
    bias = torch.zeros(H, device=dev); gamma = torch.ones(H, device=dev); beta = torch.zeros(H, device=dev)
    resid = torch.randn(M, H, device=dev)
    y = torch.empty(M, H, device=dev); yb = torch.empty(M, H, device=dev, dtype=bf); xh = torch.empty(M, H, device=dev, dtype=bf)
    rs = torch.empty(M, device=dev)
    stamps = torch.zeros(256 * 8, dtype=torch.int64, device=dev)
    g = _lib.QstGemmArgs()
    g.A, g.B, g.C, g.C2, g.bias, g.resid = A.data_ptr(), B.data_ptr(), y.data_ptr(), yb.data_ptr(), bias.data_ptr(), resid.data_ptr()
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldr, g.splits = M, H, K, K, K, H, H, 8
    e = _lib.QstLnEpi()
    e.gamma, e.beta, e.eps, e.xhat, e.rstd, e.partials = gamma.data_ptr(), beta.data_ptr(), 1e-12, xh.data_ptr(), rs.data_ptr(), stamps.data_ptr()
    for _ in range(3):
        _lib.check(lib.qst_gemm_nt_ln(g, e, 0, st))
    torch.cuda.synchronize()
    t = stamps.cpu().numpy().reshape(256, 8).astype(np.float64)
    t[:, 1] = t[:, 0]                                    # stamp 1 is no longer taken
    d = np.diff(t, axis=1)
    names = ["-", "entry -> end of K loop", "pass0 slab", "pass0 rows", "pass1 slab", "pass1 rows", "tail"]
    med = np.median(d, axis=0)
    print(f"K={K}: s_memtime ticks (100 MHz?) median per phase: " + ", ".join(f"{n} {v:.0f}" for n, v in zip(names, med)) +
          f"; total {np.median(t[:, 7] - t[:, 0]):.0f}; spread of start {t[:, 0].max() - t[:, 0].min():.0f}, of end {t[:, 7].max() - t[:, 7].min():.0f}")
