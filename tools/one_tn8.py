"""Old and 8-phase grouped wgrad kernels at one layer's shapes, a few launches each (for rocprofv3 --pmc / --kernel-trace).
   QST_H (384), QST_M (32768). (Round 4 ran it on a diagnostic build whose flush could be compiled out: QST_DIAG; that switch is gone.)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd
from quadruplet_sentence_transformer_amd import _lib

H = int(os.environ.get("QST_H", "384")); M = int(os.environ.get("QST_M", "32768")); diag = 0
I = 4 * H
lib = _lib.load(); st = _lib.current_stream_ptr()
bf = torch.bfloat16
grp = _lib.QstTnGroup(); grp.nprob = 4
keep = []
for i, (N, K) in enumerate([(H, I), (I, H), (H, H), (3 * H, H)]):
    A = torch.randn(M, N, device="cuda").to(bf); B = torch.randn(M, K, device="cuda").to(bf)
    C = torch.zeros(N, K, device="cuda"); cs = torch.zeros(N, device="cuda")
    q = grp.prob[i]
    q.A, q.B, q.C, q.colsum = A.data_ptr(), B.data_ptr(), C.data_ptr(), cs.data_ptr()
    q.M, q.N, q.K, q.lda, q.ldb, q.ldc = M, N, K, N, K, K
    keep += [A, B, C, cs]
for mode in (0, 2):
    lib.qst_gemm8_mode(mode)
    grp.splits = (diag << 16) if mode == 2 else 0
    for _ in range(6):
        _lib.check(lib.qst_gemm_tn_group(grp, st))
    torch.cuda.synchronize()
lib.qst_gemm8_mode(-1)
