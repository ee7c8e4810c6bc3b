#!/usr/bin/env python3
"""Where a plain NT GEMM launch spends its time (QstGemmArgs.splits bit 16: wall-clock stamps of wave 0 of every workgroup at
entry, after the K loop, after the epilogue; 100 MHz s_memrealtime): per workgroup the K-loop and epilogue durations, and over
the launch the timeline of when workgroups start -- for the step's K = 384 shapes."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402

lib = _lib.load()
st = _lib.current_stream_ptr()
M = 32768
dev, bf = "cuda", torch.bfloat16
for name, N, K, epi in (("QKV fwd (bf16 out)", 1152, 384, 0), ("FFN1 fwd (GELU, two outputs)", 1536, 384, 2),
                        ("GELU' dgrad", 1536, 384, 3), ("out-proj dgrad (bf16 out)", 384, 384, 0)):
    A = torch.randn(M, K, device=dev).to(bf); B = (torch.randn(N, K, device=dev) * 0.02).to(bf)
    bias = torch.zeros(N, device=dev)
    C = torch.empty(M, N, device=dev, dtype=bf); C2 = torch.empty(M, N, device=dev, dtype=bf); aux = torch.rand(M, N, device=dev).to(bf)
    ntiles = ((M + 127) // 128) * ((N + 191) // 192)
    stamps = torch.zeros(ntiles * 4, dtype=torch.int64, device=dev)
    g = _lib.QstGemmArgs()
    g.A, g.B, g.C, g.C2, g.bias, g.aux = A.data_ptr(), B.data_ptr(), C.data_ptr(), C2.data_ptr(), bias.data_ptr(), aux.data_ptr()
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc = M, N, K, K, K, N
    g.colsum, g.splits = stamps.data_ptr(), 16
    for _ in range(3):
        _lib.check(lib.qst_gemm_nt(g, epi, st))
    torch.cuda.synchronize()
    t = stamps.cpu().numpy().reshape(ntiles, 4).astype(np.float64)[:, :3] * 10.0          # ns
    t0 = t[:, 0].min()
    kl, ep = (t[:, 1] - t[:, 0]) / 1e3, (t[:, 2] - t[:, 1]) / 1e3
    start = (t[:, 0] - t0) / 1e3
    total = (t[:, 2].max() - t0) / 1e3
    order = np.sort(start)
    print(f"{name}: {ntiles} workgroups, launch {total:.1f} us; per workgroup K loop (incl. first-stage wait) median {np.median(kl):.1f} us "
          f"[p10 {np.percentile(kl, 10):.1f}, p90 {np.percentile(kl, 90):.1f}], epilogue median {np.median(ep):.1f} us "
          f"[p10 {np.percentile(ep, 10):.1f}, p90 {np.percentile(ep, 90):.1f}]; workgroup starts: 512th at {order[min(511, ntiles - 1)]:.1f} us, "
          f"last at {order[-1]:.1f} us")
