#!/usr/bin/env python3
"""Why does bench.py's configs[4] step read ~5% above tools/ab_switch.py's on every box? Same process: the trainer as
ab_switch builds it against the trainer as bench.py builds it (LR schedule arguments, four distinct batches)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd.config import PRESETS  # noqa: E402
from quadruplet_sentence_transformer_amd.synthetic import synthetic_params, synthetic_quadruplets  # noqa: E402
from quadruplet_sentence_transformer_amd.trainer import QuadrupletTrainer  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    model, B, L = "bert-base-uncased", 128, 384
    cfg = PRESETS[model]
    arena = synthetic_params(cfg, seed=14)
    def mk(bench_style):
        if bench_style:
            return QuadrupletTrainer(cfg, arena=arena, device="cuda:0", lr=2e-5, weight_decay=0.01, max_grad_norm=1.0, warmup_steps=10000,
                                     total_steps=1000000, process_group=None, world_size=1, overlap=True, use_graph=False, force_dp=False,
                                     precision="bf16", dropout=0.1, dropout_seed=14)
        return QuadrupletTrainer(cfg, arena=arena, device="cuda:0", lr=2e-5, weight_decay=0.01, max_grad_norm=1.0, dropout=(0.1, 0.1), dropout_seed=14)
    order = os.environ.get("PROBE_ORDER", "ba")          # creation order: b = bench-style arguments, a = ab_switch-style
    made = [(ch, mk(ch == "b")) for ch in order]
    one = [torch.from_numpy(x).cuda() for x in synthetic_quadruplets(cfg, B, L, seed=14)]
    four = [tuple(torch.from_numpy(x).cuda() for x in synthetic_quadruplets(cfg, B, L, seed=14, step=i, rank=0)) for i in range(4)]
    print("tokens per batch (mask sums):", int(one[1].sum()), [int(f[1].sum()) for f in four])
    k = [0]

    def step4(tr):
        tr.step(*four[k[0] % 4]); k[0] += 1
    for rnd in range(2):
        for i, (ch, tr) in enumerate(made):
            print(f"trainer #{i} ({'bench' if ch == 'b' else 'ab'}-style arguments): one batch {timed(lambda: tr.step(*one), 10):.2f} ms, "
                  f"four batches {timed(lambda: step4(tr), 12):.2f} ms", flush=True)


if __name__ == "__main__":
    main()
