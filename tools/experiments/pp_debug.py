#!/usr/bin/env python3
"""Compare the persistent NT GEMM (form 5) with the tiled kernel (form 1) element by element and print where they differ."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402


def main():
    import ctypes as C
    lib = _lib.load()
    if len(sys.argv) > 1:
        lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), sys.argv[1]))
        res, args = _lib.SIGNATURES["qst_gemm_nt"]
        lib.qst_gemm_nt.restype, lib.qst_gemm_nt.argtypes = res, args
    st = _lib.current_stream_ptr()
    bf = torch.bfloat16
    for (M, N, K) in [(256, 384, 384), (32768, 1152, 384)]:
        for epi in (0,):
            A = torch.randn(M, K, device="cuda").to(bf)
            B = (torch.randn(N, K, device="cuda") * 0.05).to(bf)
            aux = (torch.rand(M, N, device="cuda") + 0.5).to(bf)
            bias = torch.randn(N, device="cuda")
            outs = []
            for form in (1, 5, 5, 5):
                Cm = torch.full((M, N), 7.0, device="cuda", dtype=bf)
                C2 = torch.full((M, N), 7.0, device="cuda", dtype=bf)
                g = _lib.QstGemmArgs()
                g.A, g.B, g.C, g.C2, g.aux, g.bias = A.data_ptr(), B.data_ptr(), Cm.data_ptr(), C2.data_ptr(), aux.data_ptr(), bias.data_ptr()
                g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldr, g.splits = M, N, K, K, K, N, N, form
                _lib.check(lib.qst_gemm_nt(g, epi, st))
                torch.cuda.synchronize()
                outs.append(Cm.float().cpu())
            if os.environ.get("PP_ONES"):
                outs[0] = (1.0 + bias).to(bf).float().cpu().expand(M, N).contiguous()
            for k in (1, 2, 3):
                bad = (outs[k] != outs[0]).nonzero()
                msg = f"M={M} N={N} K={K} epi={epi} run{k}: {bad.shape[0]} differing elements"
                if bad.shape[0]:
                    rows = sorted(set(bad[:, 0].tolist()))
                    cols = sorted(set(bad[:, 1].tolist()))
                    d = (outs[k] - outs[0]).abs().max().item()
                    msg += f"; max |d| {d:.3g}; rows%256 {sorted(set(r % 256 for r in rows))[:24]}; cols%192 {sorted(set(c % 192 for c in cols))[:12]}"
                print(msg)


if __name__ == "__main__":
    main()
