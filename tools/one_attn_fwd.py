#!/usr/bin/env python3
"""Time attention forward at the step's shape (256 sequences x 12 heads, L = 128, d = 32), inputs cold (the timed
launches alternate between several input sets larger than the caches together)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402


def main():
    n, L, A, d = 256, 128, 12, 32
    H = A * d
    lib = _lib.load()
    st = _lib.current_stream_ptr()
    bf = torch.bfloat16
    sets = []
    for _ in range(8):
        qkv = torch.randn(n * L, 3 * H, device="cuda").to(bf)
        sets.append((qkv, torch.empty(n * L, H, dtype=bf, device="cuda")))
    mask = torch.ones(n, L, dtype=torch.int64, device="cuda")
    lse = torch.empty(n, A, L, device="cuda")

    def run(i):
        qkv, ctx = sets[i % len(sets)]
        _lib.check(lib.qst_attention_fwd(qkv.data_ptr(), mask.data_ptr(), None, n, L, A, d, ctx.data_ptr(), lse.data_ptr(), st))
    for i in range(8):
        run(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 40
    for i in range(reps):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    print(f"attention forward n={n} L={L} A={A} d={d}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us")


if __name__ == "__main__":
    main()
