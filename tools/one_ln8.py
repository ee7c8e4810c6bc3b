"""A few launches of the H = 768 GEMM + LayerNorm kernel (forward K = 3072 and K = 768, backward K = 3072) and of the pair it
replaces at M = 196,608, for the counter passes of tools/pmc_quick.sh (bash tools/pmc_quick.sh ln8 tools/one_ln8.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402

lib = _lib.load()
st = _lib.current_stream_ptr()
bf = torch.bfloat16
M, N = int(os.environ.get("QST_M", "196608")), 768


def gargs(**kw):
    g = _lib.QstGemmArgs()
    for k, v in kw.items():
        setattr(g, k, v.data_ptr() if torch.is_tensor(v) else v)
    return g


gamma = torch.ones(N, device="cuda"); beta = torch.zeros(N, device="cuda"); bias = torch.randn(N, device="cuda")
resid = torch.randn(M, N, device="cuda"); s = torch.empty(M, N, device="cuda")
y = torch.empty(M, N, device="cuda"); yb = torch.empty(M, N, device="cuda", dtype=bf)
xh = torch.empty(M, N, device="cuda", dtype=bf); rs = torch.empty(M, device="cuda")
part = torch.zeros((M + 127) // 128, 2, N, device="cuda")
dg = torch.zeros(N, device="cuda"); db = torch.zeros(N, device="cuda")
scratch = torch.empty(lib.qst_ln_bwd_scratch_bytes(M, N) // 4, device="cuda")
e0 = _lib.QstLnEpi(); e0.gamma, e0.beta, e0.eps, e0.xhat, e0.rstd = gamma.data_ptr(), beta.data_ptr(), 1e-12, xh.data_ptr(), rs.data_ptr()
e1 = _lib.QstLnEpi(); e1.gamma, e1.xhat, e1.rstd, e1.partials = gamma.data_ptr(), xh.data_ptr(), rs.data_ptr(), part.data_ptr()
for K in (3072, 768):
    A = torch.randn(M, K, device="cuda").to(bf); B = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
    for _ in range(3):
        _lib.check(lib.qst_gemm_nt_ln(gargs(A=A, B=B, C=y, C2=yb, bias=bias, resid=resid, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, ldr=N), e0, 0, st))
        if K == 3072:
            _lib.check(lib.qst_gemm_nt_ln(gargs(A=A, B=B, C=y, C2=yb, resid=resid, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, ldr=N), e1, 1, st))
            _lib.check(lib.qst_gemm_nt(gargs(A=A, B=B, C=s, bias=bias, resid=resid, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, ldr=N), 1, st))
            _lib.check(lib.qst_ln_fwd(s.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-12, M, N, y.data_ptr(), yb.data_ptr(), xh.data_ptr(),
                                      rs.data_ptr(), st))
            _lib.check(lib.qst_ln_bwd(s.data_ptr(), xh.data_ptr(), rs.data_ptr(), gamma.data_ptr(), M, N, y.data_ptr(), yb.data_ptr(),
                                      dg.data_ptr(), db.data_ptr(), scratch.data_ptr(), st))
    torch.cuda.synchronize()
