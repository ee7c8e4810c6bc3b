#!/usr/bin/env python3
"""Timing experiments on the fused feed-forward kernel (csrc/ffn.hip) at M = 32768: QstFfnArgs.diag, shown as the bits
1 = drop the activation-panel loads, 2 = drop the weight loads (zero-record descriptors: the instruction stream, waits and
barriers stay, the memory traffic goes), 4 = L2 touch-ahead. Outputs of the drop builds are wrong by construction."""
import ctypes as C
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
    M, H, I = int(os.environ.get("M", 32768)), 384, 1536
    dev, bf = "cuda", torch.bfloat16
    A = torch.randn(M, H, device=dev).to(bf)
    W1 = (torch.randn(I, H, device=dev) * 0.02).to(bf); W2 = (torch.randn(H, I, device=dev) * 0.02).to(bf)
    b1 = torch.zeros(I, device=dev); b2 = torch.zeros(H, device=dev); gamma = torch.ones(H, device=dev); beta = torch.zeros(H, device=dev)
    resid = torch.randn(M, H, device=dev)
    gp = torch.rand(M, I, device=dev).to(bf); hh = torch.empty(M, I, device=dev, dtype=bf); du = torch.empty(M, I, device=dev, dtype=bf)
    y = torch.empty(M, H, device=dev); yb = torch.empty(M, H, device=dev, dtype=bf); xh = torch.randn(M, H, device=dev).to(bf)
    rs = torch.rand(M, device=dev) + 0.5
    part = torch.empty((M + 127) // 128, 2, H, device=dev)
    e = _lib.QstLnEpi()
    e.gamma, e.beta, e.eps, e.xhat, e.rstd, e.partials = gamma.data_ptr(), beta.data_ptr(), 1e-12, xh.data_ptr(), rs.data_ptr(), part.data_ptr()

    def args(mode, save):
        f = _lib.QstFfnArgs()
        f.A, f.B1, f.B2, f.resid, f.C, f.C2, f.M, f.H, f.I = A.data_ptr(), W1.data_ptr(), W2.data_ptr(), resid.data_ptr(), y.data_ptr(), yb.data_ptr(), M, H, I
        if mode == 0:
            f.bias1, f.bias2 = b1.data_ptr(), b2.data_ptr()
            if save:
                f.save_gp, f.save_h = gp.data_ptr(), hh.data_ptr()
        else:
            f.aux, f.save_h = gp.data_ptr(), du.data_ptr()
        return f
    cases = [("fwd inference", args(0, False), 0), ("fwd training", args(0, True), 0), ("bwd", args(1, True), 1)]
    fl = 4.0 * M * H * I
    for bits in (0, 4, 1, 2, 3, 7):
        row = []
        for name, f, mode in cases:
            f.diag = (bits & 3) | (0 if bits & 4 else 4)          # QstFfnArgs.diag: bit 2 switches the touch-ahead OFF
            best = min(timeit(lambda: _lib.check(lib.qst_ffn_chain(f, e, mode, st))) for _ in range(3))
            row.append(f"{name} {best:7.1f} us ({fl / best / 1e6:6.1f} TF)")
        print(f"diag={bits}: " + "   ".join(row))


if __name__ == "__main__":
    main()
