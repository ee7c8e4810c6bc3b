#!/usr/bin/env python3
"""Time attention backward at the step's shape: single-workgroup kernel vs the two-kernel path."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    n, L, A, d = 256, 128, 12, 32
    H = A * d
    lib = _lib.load()
    st = _lib.current_stream_ptr()
    bf = torch.bfloat16
    qkv = torch.randn(n * L, 3 * H, device="cuda").to(bf)
    mask = torch.ones(n, L, dtype=torch.int64, device="cuda")
    ctx = torch.empty(n * L, H, dtype=bf, device="cuda")
    lse = torch.empty(n, A, L, device="cuda")
    _lib.check(lib.qst_attention_fwd(qkv.data_ptr(), mask.data_ptr(), None, n, L, A, d, ctx.data_ptr(), lse.data_ptr(), st))
    dctx = torch.randn(n * L, H, device="cuda").to(bf)
    dq = torch.empty(n * L, 3 * H, dtype=bf, device="cuda")
    delta = torch.empty(n, A, L, device="cuda")

    def run():
        _lib.check(lib.qst_attention_bwd(qkv.data_ptr(), ctx.data_ptr(), dctx.data_ptr(), lse.data_ptr(), mask.data_ptr(),
                                         None, n, L, A, d, dq.data_ptr(), None, delta.data_ptr(), st))
    q = _lib.QstAttnDesc()
    q.qkv, q.mask, q.nseq, q.L, q.A, q.d = qkv.data_ptr(), mask.data_ptr(), n, L, A, d
    q.ctx, q.lse, q.dctx, q.dqkv, q.delta_scratch = ctx.data_ptr(), lse.data_ptr(), dctx.data_ptr(), dq.data_ptr(), delta.data_ptr()
    q.force_split = 1
    t1 = timeit(run)
    t2 = timeit(lambda: _lib.check(lib.qst_attention_bwd_ex(q, st)))
    print(f"attention backward n={n} L={L} A={A} d={d}: single-workgroup {t1:.1f} us, two-kernel {t2:.1f} us")


if __name__ == "__main__":
    main()
