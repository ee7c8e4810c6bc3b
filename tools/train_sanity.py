#!/usr/bin/env python3
"""End-to-end sanity at BASELINE configs[1] dims: a few hundred full steps on four fixed synthetic batches must drive the
quadruplet loss down and keep every parameter finite -- in every training precision, from the same initial parameters,
on the same batches, dropout off, so the trajectories are comparable: `bf16x3` is the fp32-class path (the reference trains
in fp32, training/main.py:142), the others are measured against it step by step.

    python tools/train_sanity.py [bf16 f16 f16w fp8 bf16x3] [steps=300]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd.config import PRESETS  # noqa: E402
from quadruplet_sentence_transformer_amd.synthetic import synthetic_params, synthetic_quadruplets  # noqa: E402
from quadruplet_sentence_transformer_amd.trainer import QuadrupletTrainer  # noqa: E402

ALL = ("bf16", "f16", "f16w", "fp8", "bf16x3")


def run(cfg, prec, batches, steps):
    tr = QuadrupletTrainer(cfg, arena=synthetic_params(cfg, seed=14), device="cuda:0", lr=5e-5, weight_decay=0.01,
                           max_grad_norm=1.0, warmup_steps=20, total_steps=steps + 100, precision=prec, dropout=None)
    losses = []
    for step in range(steps):
        losses.append(tr.step(*batches[step % 4]))
    losses = torch.cat([x.reshape(1).float() for x in losses]).cpu()
    assert torch.isfinite(losses).all() and torch.isfinite(tr.enc.params).all(), prec
    assert losses[-4:].mean() < losses[:4].mean() - 0.2, f"{prec}: training did not reduce the loss"
    return losses, tr.enc.params.detach().float().cpu().reshape(-1).clone(), float(tr.enc.grad_norm)


def main():
    a = sys.argv[1:]
    steps = next((int(x.split("=")[1]) for x in a if x.startswith("steps=")), 300)
    precs = [x for x in a if x in ALL] or ["bf16"]
    cfg = PRESETS["all-MiniLM-L6-v2"]
    batches = [tuple(torch.from_numpy(x).cuda() for x in synthetic_quadruplets(cfg, 64, 128, seed=14, step=i)) for i in range(4)]
    out = {p: run(cfg, p, batches, steps) for p in precs}
    marks = sorted(set([0, 1, 2, 3] + list(range(24, steps, 25)) + [steps - 1]))
    print(f"all-MiniLM-L6-v2 dims, 64 quadruplets x 128 tokens, four fixed batches in turn, {steps} steps, lr 5e-5 (20 warm-up steps), "
          "dropout off, same initial parameters")
    print("step   " + "".join(f"{p:>10s}" for p in precs))
    for m in marks:
        print(f"{m:5d}  " + "".join(f"{out[p][0][m].item():10.4f}" for p in precs))
    ref = "bf16x3" if "bf16x3" in out else None
    for p in precs:
        l, w, gn = out[p]
        line = f"{p}: loss {l[:4].mean():.4f} -> {l[-4:].mean():.4f}, final grad norm {gn:.4f}"
        if ref and p != ref:
            lr_, wr = out[ref][0], out[ref][1]
            w0 = torch.from_numpy(synthetic_params(cfg, seed=14)).reshape(-1)
            line += (f"; against {ref}: max |d loss| over the run {float((l - lr_).abs().max()):.2e}, "
                     f"end-point distance |w - w_{ref}| = {float((w - wr).norm() / (wr - w0).norm()):.2e} of the distance the {ref} "
                     f"run travelled (|w_{ref} - w_0| = {float((wr - w0).norm()):.3f})")
        print(line)
    print("ok")


if __name__ == "__main__":
    main()
