"""Weight gradients at parity precision (qst_gemm_tn_x3): time against the number of workgroups sharing the token reduction.
    python tools/x3_wgrad_sweep.py [H] [M]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402

lib = _lib.load()
H = int(sys.argv[1]) if len(sys.argv) > 1 else 384
M = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
I = 4 * H
st = _lib.current_stream_ptr()
for out, inn in [(3 * H, H), (H, H), (I, H), (H, I)]:
    dY, X = torch.randn(M, out, device="cuda"), torch.randn(M, inn, device="cuda")
    C, cs = torch.zeros(out, inn, device="cuda"), torch.zeros(out, device="cuda")
    line = f"dW [{out}, {inn}] over {M} rows:"
    for target in (128, 256, 512, 1024, 2048):
        g = _lib.QstGemmArgs()
        g.A, g.B, g.C, g.colsum = dY.data_ptr(), X.data_ptr(), C.data_ptr(), cs.data_ptr()
        g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.splits = M, out, inn, out, inn, inn, target
        for _ in range(2):
            _lib.check(lib.qst_gemm_tn_x3(g, st))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            _lib.check(lib.qst_gemm_tn_x3(g, st))
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        line += f"  {target}: {us:.0f} us ({6.0 * M * out * inn / us * 1e-6:.0f} TF/s of MFMA)"
    print(line)
