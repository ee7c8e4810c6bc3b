"""Launch the grouped wgrad kernel at BASELINE configs[1] shapes (for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd
from quadruplet_sentence_transformer_amd import _lib


def make_group(M=32768, H=384, I=1536, dev="cuda"):
    bf = torch.bfloat16
    keep = []
    grp = _lib.QstTnGroup()
    grp.nprob, grp.splits = 4, int(__import__("os").environ.get("QST_SPLITS", "0"))
    for i, (N, K) in enumerate([(H, I), (I, H), (H, H), (3 * H, H)]):
        A = torch.randn(M, N, device=dev).to(bf); B = torch.randn(M, K, device=dev).to(bf)
        C = torch.zeros(N, K, device=dev); cs = torch.zeros(N, device=dev)
        q = grp.prob[i]
        q.A, q.B, q.C, q.colsum = A.data_ptr(), B.data_ptr(), C.data_ptr(), cs.data_ptr()
        q.M, q.N, q.K, q.lda, q.ldb, q.ldc = M, N, K, N, K, K
        keep += [A, B, C, cs]
    flops = 2.0 * M * (H * I + I * H + H * H + 3 * H * H)
    return grp, keep, flops


if __name__ == "__main__":
    lib = _lib.load(); st = _lib.current_stream_ptr()
    grp, keep, flops = make_group(M=int(os.environ.get("QST_M", "16384")))
    for _ in range(5):
        _lib.check(lib.qst_gemm_tn_group(grp, st))
    torch.cuda.synchronize()
