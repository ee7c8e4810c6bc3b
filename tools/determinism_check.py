#!/usr/bin/env python3
"""Run the same forward + backward several times on identical inputs and report, per parameter tensor, the largest relative
L2 difference between runs (fp32 atomics in the wgrad flush reorder sums: ~1e-7 expected; anything larger is a race).
The parity path (precision bf16x3) accumulates dW, the fused bias sums and the column sums with fp32 atomics across K-shares and
row chunks since round 4, so its gradients are reproducible to fp32 summation order only, like the bf16 path's:
pass the precision to see the spread (measured round 5: see DESIGN.md section 2).
usage: determinism_check.py [model] [B] [L] [runs] [precision: bf16 | f16 | f16w | bf16x3]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd.config import PRESETS, build_layout  # noqa: E402
from quadruplet_sentence_transformer_amd.encoder import HipEncoder  # noqa: E402
from quadruplet_sentence_transformer_amd.synthetic import synthetic_params, synthetic_quadruplets  # noqa: E402


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "all-MiniLM-L6-v2"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    runs = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    prec = sys.argv[5] if len(sys.argv) > 5 else "bf16"
    cfg = PRESETS[model]
    enc = HipEncoder(cfg)
    enc.load_arena(synthetic_params(cfg, seed=14, std=0.04, bias_std=0.02, ln_jitter=0.05))
    enc.ensure_train_state()
    ids, mask, types = [torch.from_numpy(x).view(4 * B, L).cuda() for x in synthetic_quadruplets(cfg, B, L, seed=14, ragged=True)]
    types = types if cfg.type_vocab_size else None
    g = torch.randn(4 * B, cfg.hidden_size, generator=torch.Generator().manual_seed(1)).cuda()
    outs = []
    for _ in range(runs):
        emb, _, saved = enc.forward(ids, mask, types, training=True, precision=prec)
        enc.grads.zero_()
        enc.backward(ids, mask, types, g, saved, precision=prec)
        torch.cuda.synchronize()
        outs.append((emb.clone(), enc.grads.clone()))
    segs, _ = build_layout(cfg)
    print("precision", prec)
    print("embeddings identical:", all(torch.equal(outs[0][0], o[0]) for o in outs[1:]))
    worst = []
    for s in segs:
        ref = outs[0][1][s.offset:s.offset + s.numel]
        d = max(((o[1][s.offset:s.offset + s.numel] - ref).norm() / ref.norm().clamp_min(1e-30)).item() for o in outs[1:])
        worst.append((d, s.name))
    worst.sort(reverse=True)
    for d, n in worst[:8]:
        print(f"{n:24s} max relative L2 difference between runs {d:.3e}")


if __name__ == "__main__":
    main()
