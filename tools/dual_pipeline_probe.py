#!/usr/bin/env python3
"""Potential of two half-chip pipelines: two independent half-batch training steps (B = 32 quadruplets each) on two
streams created with hipExtStreamCreateWithCUMask (CU-mask bits 0-127 / 128-255 = 16 CUs of every XCD each), against
one full-batch step (B = 64) on the whole chip. Prototype only: each half has its own parameters and optimiser here."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd.config import PRESETS  # noqa: E402
from quadruplet_sentence_transformer_amd.synthetic import synthetic_params, synthetic_quadruplets  # noqa: E402
from quadruplet_sentence_transformer_amd.trainer import QuadrupletTrainer  # noqa: E402


def masked_stream(lo, hi):
    hip = C.CDLL("libamdhip64.so")
    words = (C.c_uint32 * 8)()
    for b in range(lo, hi):
        words[b // 32] |= 1 << (b % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def main():
    cfg = PRESETS["all-MiniLM-L6-v2"]
    arena = synthetic_params(cfg, seed=14)
    L = 128
    full = QuadrupletTrainer(cfg, arena=arena, device="cuda:0")
    bf = [torch.from_numpy(x).cuda() for x in synthetic_quadruplets(cfg, 64, L, seed=14)]
    for _ in range(5):
        full.step(*bf)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        full.step(*bf)
    torch.cuda.synchronize()
    t_full = (time.perf_counter() - t0) / 20
    print(f"one B=64 step on the whole chip: {t_full * 1e3:.3f} ms")

    halves = [QuadrupletTrainer(cfg, arena=arena, device="cuda:0") for _ in range(2)]
    bh = [[torch.from_numpy(x).cuda() for x in synthetic_quadruplets(cfg, 32, L, seed=14, step=i)] for i in range(2)]
    for mode, streams in (("two plain streams", [torch.cuda.Stream(), torch.cuda.Stream()]),
                          ("two CU-masked streams (128 CUs each)", [masked_stream(0, 128), masked_stream(128, 256)]),
                          ("one stream, halves back to back", [torch.cuda.current_stream()] * 2)):
        def both():
            for tr, b, st in zip(halves, bh, streams):
                with torch.cuda.stream(st):
                    tr.step(*b)
        for _ in range(5):
            both()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            both()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 20
        print(f"two B=32 steps, {mode}: {t * 1e3:.3f} ms  ({t / t_full:.3f} x the full-batch step; each half also runs its own AdamW, ~0.12 ms)")


if __name__ == "__main__":
    main()
