"""Quick correctness check of qst_gemm_nt forms against a torch fp32 matmul on the device (tools; the tests use the CPU oracle)."""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from quadruplet_sentence_transformer_amd import _lib
lib = _lib.load()
st = _lib.current_stream_ptr
def args(**kw):
    g = _lib.QstGemmArgs()
    g._keep = [v for v in kw.values() if torch.is_tensor(v)]
    for k, v in kw.items():
        setattr(g, k, v.data_ptr() if torch.is_tensor(v) else v)
    return g
g = torch.Generator(device="cuda").manual_seed(1)
for (M, N, K) in [(256, 384, 128), (1000, 768, 768), (4096, 768, 448)]:
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    B = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    ref = A.float() @ B.float().t() + bias
    for form in (0x80, 0x20, 0x40):
        C = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        _lib.check(lib.qst_gemm_nt(args(A=A, B=B, C=C, bias=bias, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, splits=form), 0, st()))
        d = (C.float() - ref).abs()
        bad = (d > 0.05 + 0.01 * ref.abs())
        print(f"M={M} N={N} K={K} form {form:#x}: max|d| {d.max().item():.3g}, bad {int(bad.sum())} of {M*N}")
        if bad.any():
            idx = bad.nonzero()
            print("   first bad (m, n):", idx[:6].tolist(), " rows hit:", idx[:, 0].unique().numel(), "cols hit:", idx[:, 1].unique().numel())
            r, c = idx[0].tolist()
            print("   got", C[r, c:c+8].float().tolist(), "\n   ref", ref[r, c:c+8].tolist())
