#!/usr/bin/env python3
"""Time libqst's retrieval scoring (normalise + split-bf16 x3 GEMM + radix-select top-k) at the reference's evaluation
shape (corpus_chunk_size 50000, training/main.py:178) against torch (cos_sim + topk), yardstick only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import util  # noqa: E402


def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    for nq, nc, dim, k in [(1000, 50000, 384, 100), (4096, 50000, 768, 100)]:
        q = torch.randn(nq, dim, device="cuda")
        c = torch.randn(nc, dim, device="cuda")
        t_qst = timeit(lambda: util.topk_scores(q, c, k, cosine=True))
        t_torch = timeit(lambda: torch.topk(util.cos_sim(q, c), k, dim=1))
        print(f"nq={nq} nc={nc} dim={dim} k={k}: libqst {t_qst:.2f} ms   torch fp32 matmul + topk {t_torch:.2f} ms")


if __name__ == "__main__":
    main()
