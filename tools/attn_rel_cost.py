#!/usr/bin/env python3
"""What the MPNet relative-position bias costs the attention kernels at BASELINE configs[2] (128 seq x 12 heads,
L = 256, d = 64): forward / backward with and without the bias (and its gradient)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    n, L, A, d = 128, 256, 12, 64
    H = A * d
    lib = _lib.load()
    st = _lib.current_stream_ptr()
    bf = torch.bfloat16
    qkv = torch.randn(n * L, 3 * H, device="cuda").to(bf)
    mask = torch.ones(n, L, dtype=torch.int64, device="cuda")
    rel = torch.randn(A, 2 * L, device="cuda") * 0.1          # relative-position vectors
    drel = torch.zeros(A, 2 * L, device="cuda")
    ctx = torch.empty(n * L, H, dtype=bf, device="cuda")
    lse = torch.empty(n, A, L, device="cuda")
    dctx = torch.randn(n * L, H, device="cuda").to(bf)
    dq = torch.empty(n * L, 3 * H, dtype=bf, device="cuda")
    delta = torch.empty(n, A, L, device="cuda")
    for name, r, dr in [("no bias", None, None), ("bias, no bias gradient", rel, None), ("bias + gradient", rel, drel)]:
        rp = None if r is None else r.data_ptr()
        dp = None if dr is None else dr.data_ptr()
        tf = timeit(lambda: _lib.check(lib.qst_attention_fwd(qkv.data_ptr(), mask.data_ptr(), rp, n, L, A, d, ctx.data_ptr(), lse.data_ptr(), st)))
        tb = timeit(lambda: _lib.check(lib.qst_attention_bwd(qkv.data_ptr(), ctx.data_ptr(), dctx.data_ptr(), lse.data_ptr(), mask.data_ptr(), rp, n, L, A, d, dq.data_ptr(), dp, delta.data_ptr(), st)))
        print(f"{name:26s} forward {tf:7.1f} us   backward {tb:7.1f} us")


if __name__ == "__main__":
    main()
