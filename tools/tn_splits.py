import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd
from quadruplet_sentence_transformer_amd import _lib
from tools.gemm_bench import timeit
lib = _lib.load(); st = _lib.current_stream_ptr(); dev="cuda"; bf=torch.bfloat16
M=32768; H=384; I=1536
for name, N, K in [("dW2 [H,I]", H, I), ("dW1 [I,H]", I, H), ("dWo [H,H]", H, H), ("dWqkv [3H,H]", 3 * H, H)]:
    A = torch.randn(M, N, device=dev).to(bf); B = torch.randn(M, K, device=dev).to(bf)
    C = torch.zeros(N, K, device=dev); cs = torch.zeros(N, device=dev)
    for sp in (8, 16, 24, 32, 48, 64):
        g = _lib.QstGemmArgs(); g.A, g.B, g.C, g.colsum = A.data_ptr(), B.data_ptr(), C.data_ptr(), cs.data_ptr()
        g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.splits = M, N, K, N, K, K, sp
        us = timeit(lambda: _lib.check(lib.qst_gemm_tn(g, st)))
        print(f"{name:14s} splits={sp:3d} {us:8.1f} us {2.0*M*N*K/us/1e6:8.1f} TF")
