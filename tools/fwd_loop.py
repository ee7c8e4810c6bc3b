#!/usr/bin/env python3
"""Forward-only loop for profiling: python tools/fwd_loop.py <model> <batch quadruplets> <seq_len> <precision> [iters]
(under rocprofv3 --kernel-trace --stats for the per-kernel breakdown of an inference precision mode)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd.config import PRESETS  # noqa: E402
from quadruplet_sentence_transformer_amd.synthetic import synthetic_params, synthetic_quadruplets  # noqa: E402
from quadruplet_sentence_transformer_amd.trainer import QuadrupletTrainer  # noqa: E402


def main():
    model, B, L, prec = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    cfg = PRESETS[model]
    tr = QuadrupletTrainer(cfg, arena=synthetic_params(cfg, seed=14), device="cuda:0")
    batch = [torch.from_numpy(x).cuda() for x in synthetic_quadruplets(cfg, B, L, seed=14)]
    for _ in range(2):
        tr.forward_loss(*batch, precision=prec)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        tr.forward_loss(*batch, precision=prec)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"{model} B={B} L={L} {prec}: {dt * 1e3:.2f} ms per forward, {B / dt:.1f} quadruplets/s")


if __name__ == "__main__":
    main()
