#!/usr/bin/env python3
"""Does running the two halves of a batch on two streams de-phase the HBM-bound epilogues from the K loops?
One fused GEMM+LayerNorm launch at M = 32768 against two M = 16384 launches issued on two streams, and a chain of
four dependent launches per stream (out+LN1 -> FFN1 -> FFN2+LN2 -> QKV) against the same chain at full M."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402

lib = _lib.load()
bf = torch.bfloat16
H, I = 384, 1536


def mk(M):
    d = {}
    d["ctx"] = torch.randn(M, H, device="cuda").to(bf)
    d["x"] = torch.randn(M, H, device="cuda")
    d["y1"] = torch.empty(M, H, device="cuda"); d["y1b"] = torch.empty(M, H, device="cuda", dtype=bf)
    d["xh"] = torch.empty(M, H, device="cuda", dtype=bf); d["rs"] = torch.empty(M, device="cuda")
    d["u"] = torch.empty(M, I, device="cuda", dtype=bf); d["h"] = torch.empty(M, I, device="cuda", dtype=bf)
    d["x2"] = torch.empty(M, H, device="cuda"); d["x2b"] = torch.empty(M, H, device="cuda", dtype=bf)
    d["qkv"] = torch.empty(M, 3 * H, device="cuda", dtype=bf)
    return d


W = {k: (torch.randn(*s, device="cuda") * 0.02).to(bf) for k, s in
     dict(o=(H, H), w1=(I, H), w2=(H, I), qkv=(3 * H, H)).items()}
bias = {k: torch.zeros(n, device="cuda") for k, n in dict(o=H, w1=I, w2=H, qkv=3 * H).items()}
gamma, beta = torch.ones(H, device="cuda"), torch.zeros(H, device="cuda")


def gargs(A, B, C, C2, b, resid, M, N, K):
    g = _lib.QstGemmArgs()
    g.A, g.B, g.C, g.C2, g.bias = A.data_ptr(), B.data_ptr(), C.data_ptr(), (C2.data_ptr() if C2 is not None else None), b.data_ptr()
    g.resid = resid.data_ptr() if resid is not None else None
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldr = M, N, K, K, K, N, N
    return g


def chain(d, M, st):
    e = _lib.QstLnEpi()
    e.gamma, e.beta, e.eps, e.xhat, e.rstd = gamma.data_ptr(), beta.data_ptr(), 1e-12, d["xh"].data_ptr(), d["rs"].data_ptr()
    _lib.check(lib.qst_gemm_nt_ln(gargs(d["ctx"], W["o"], d["y1"], d["y1b"], bias["o"], d["x"], M, H, H), e, 0, st))
    _lib.check(lib.qst_gemm_nt(gargs(d["y1b"], W["w1"], d["u"], d["h"], bias["w1"], None, M, I, H), 2, st))
    _lib.check(lib.qst_gemm_nt_ln(gargs(d["h"], W["w2"], d["x2"], d["x2b"], bias["w2"], d["y1"], M, H, I), e, 0, st))
    _lib.check(lib.qst_gemm_nt(gargs(d["x2b"], W["qkv"], d["qkv"], None, bias["qkv"], None, M, 3 * H, H), 0, st))


def main():
    M = 32768
    full = mk(M)
    halves = [mk(M // 2), mk(M // 2)]
    s0 = torch.cuda.current_stream()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def one():
        chain(full, M, s0.cuda_stream)

    def two():
        s1.wait_stream(s0); s2.wait_stream(s0)
        chain(halves[0], M // 2, s1.cuda_stream)
        chain(halves[1], M // 2, s2.cuda_stream)
        s0.wait_stream(s1); s0.wait_stream(s2)

    for name, fn in (("one stream, M = 32768", one), ("two streams, M = 16384 each", two)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:32s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us per 4-GEMM chain")


if __name__ == "__main__":
    main()
