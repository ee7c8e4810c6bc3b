#!/usr/bin/env python3
"""Same-process A/B of an encoder-handle option on the training step and the forward-only pass:

    python tools/ab_switch.py set_ffn_chain 1 7 [model] [batch] [seq_len] [rounds]

Calls HipEncoder.<option>(v) for each value v in turn, several rounds, and prints ms per step for both; the two settings alternate in
one process on one box, so clock / box differences cancel (boxes of the pool differ by +-4%)."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402
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
    name, v0, v1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    model = sys.argv[4] if len(sys.argv) > 4 else "all-MiniLM-L6-v2"
    B = int(sys.argv[5]) if len(sys.argv) > 5 else 64
    L = int(sys.argv[6]) if len(sys.argv) > 6 else 128
    rounds = int(sys.argv[7]) if len(sys.argv) > 7 else 4
    cfg = PRESETS[model]
    drop = float(os.environ.get("AB_DROPOUT", "0"))          # AB_DROPOUT=0.1: the reference's train()-mode step, as bench.py times it
    # (the LR schedule of bench.py: 10,000 warm-up steps. Without it the trainer overfits its one batch within tens of steps,
    #  the hinges go inactive, every gradient is exactly zero -- and a power-limited chip multiplies zeros 4-5% faster:
    #  tools/step_timing_probe.py)
    tr = QuadrupletTrainer(cfg, arena=synthetic_params(cfg, seed=14), device="cuda:0", lr=2e-5, weight_decay=0.01, max_grad_norm=1.0,
                           warmup_steps=10000, total_steps=1000000,
                           dropout=(drop, drop) if drop > 0 else None, dropout_seed=14,
                           precision=os.environ.get("AB_PRECISION", "bf16"))      # AB_PRECISION=fp8 | f16 | f16w | bf16x3
    batch = [torch.from_numpy(x).cuda() for x in synthetic_quadruplets(cfg, B, L, seed=14)]
    sw = getattr(tr.enc.lib, name[4:]) if name.startswith("lib:") else getattr(tr.enc, name)     # lib:qst_gemm8_stagger = a process-wide knob
    res = {v0: [[], []], v1: [[], []]}
    for _ in range(rounds):
        for v in (v0, v1):
            sw(v)
            res[v][0].append(timed(lambda: tr.step(*batch), 20))
            res[v][1].append(timed(lambda: tr.forward_loss(*batch), 20))
    for v in (v0, v1):
        st, fw = res[v]
        print(f"{name}({v}): step {min(st):.3f} ms (runs {' '.join(f'{x:.3f}' for x in st)})   "
              f"forward-only {min(fw):.3f} ms (runs {' '.join(f'{x:.3f}' for x in fw)})")
    sw(v0)


if __name__ == "__main__":
    main()
