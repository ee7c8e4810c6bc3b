#!/usr/bin/env python3
"""Forward rate of the three precisions over the same fp32 arena (MiniLM, 256 sequences x 128 tokens)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd
from quadruplet_sentence_transformer_amd.config import PRESETS
from quadruplet_sentence_transformer_amd.encoder import HipEncoder
from quadruplet_sentence_transformer_amd.synthetic import synthetic_params, synthetic_quadruplets
cfg = PRESETS["all-MiniLM-L6-v2"]
enc = HipEncoder(cfg); enc.load_arena(synthetic_params(cfg, seed=14))
ids, mask, types = synthetic_quadruplets(cfg, 64, 128, seed=14)
i, m, t = (torch.from_numpy(x).view(256, 128).cuda() for x in (ids, mask, types))
for prec in ("bf16", "fp8", "bf16x3"):
    for _ in range(3): enc.forward(i, m, t, precision=prec)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): enc.forward(i, m, t, precision=prec)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"{prec:7s} forward of 256 x 128 tokens: {dt*1e3:.2f} ms  ({64/dt:.0f} quadruplets/s)")
