#!/usr/bin/env python3
"""End-to-end sanity at BASELINE configs[1] dims: a few hundred full steps on four fixed synthetic batches must drive the
quadruplet loss down and keep every parameter finite."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd.config import PRESETS  # noqa: E402
from quadruplet_sentence_transformer_amd.synthetic import synthetic_params, synthetic_quadruplets  # noqa: E402
from quadruplet_sentence_transformer_amd.trainer import QuadrupletTrainer  # noqa: E402


def main():
    cfg = PRESETS["all-MiniLM-L6-v2"]
    tr = QuadrupletTrainer(cfg, arena=synthetic_params(cfg, seed=14), device="cuda:0", lr=5e-5, weight_decay=0.01,
                           max_grad_norm=1.0, warmup_steps=20, total_steps=400)
    batches = [tuple(torch.from_numpy(x).cuda() for x in synthetic_quadruplets(cfg, 64, 128, seed=14, step=i)) for i in range(4)]
    losses = []
    for step in range(300):
        losses.append(tr.step(*batches[step % 4]))
    losses = torch.cat(losses).cpu()
    print("loss: first 4 steps", [round(v, 4) for v in losses[:4].tolist()], " last 4 steps", [round(v, 4) for v in losses[-4:].tolist()])
    assert torch.isfinite(losses).all() and torch.isfinite(tr.enc.params).all()
    assert losses[-4:].mean() < losses[:4].mean() - 0.2, "training did not reduce the loss"
    print("ok: grad norm at the end", float(tr.enc.grad_norm))


if __name__ == "__main__":
    main()
