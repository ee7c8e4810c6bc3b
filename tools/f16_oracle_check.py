"""CPU: the oracle with bf16- and f16-rounded matrix-core operands against the committed HF golden vectors (round 5, the
experiment of VERDICT r04 'missing #1'): per golden case max|d emb|, whether it is inside rtol 1e-3 / atol 1e-4, |d loss|,
and the worst per-tensor relative L2 gradient error. The f16 backward runs under a power-of-two loss scale, as the kernels'
does (grad_emb * S in, 1/S out): `--scale`. Usage: python tools/f16_oracle_check.py [--scale 1024] [--cases a,b]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd.config import PRESETS, build_layout  # noqa: E402
from quadruplet_sentence_transformer_amd.synthetic import synthetic_params  # noqa: E402
from oracle import torch_ref as R  # noqa: E402
from tests.test_oracle_golden import CLI, ENC_CASES, golden_inputs  # noqa: E402

CASES = ENC_CASES + [("minilm_l128", "all-MiniLM-L6-v2", 2, 128, dict(std=0.02), "norms")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1024.0)
    ap.add_argument("--cases", default="")
    ap.add_argument("--no-grads", action="store_true")
    a = ap.parse_args()
    g = np.load(os.path.join(ROOT, "tests", "golden", "encoder_golden.npz"))
    print(f"{'case':20s} {'op':5s} {'max|d emb|':>11s} {'in tol':>6s} {'|d loss|':>9s} {'worst grad relL2':>17s}")
    for key, preset, B, L, wkw, store in CASES:
        if a.cases and key not in a.cases.split(","):
            continue
        cfg = PRESETS[preset]
        arena = synthetic_params(cfg, seed=14, **wkw)
        ids, mask, types = golden_inputs(key, cfg, B, L)
        segs, total = build_layout(cfg)
        for op in ("bf16", "f16"):
            P = R.arena_to_dict(arena, cfg, requires_grad=not a.no_grads)
            loss, emb = R.quadruplet_step(P, cfg, torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(types), CLI,
                                          bf16_operands=op)
            e, ref = emb.detach().numpy(), g[key + "_emb"]
            d = np.abs(e - ref)
            ok = bool((d <= 1e-4 + 1e-3 * np.abs(ref)).all())
            worst = float("nan")
            if not a.no_grads:
                S = a.scale if op == "f16" else 1.0
                (loss * S).backward()
                ga = np.zeros(total, np.float32)
                for s in segs:
                    ga[s.offset:s.offset + s.numel] = P[s.name].grad.numpy().reshape(-1) / S
                if store == "full":
                    refg = g[key + "_grads"]
                    worst = max(np.linalg.norm(ga[s.offset:s.offset + s.numel] - refg[s.offset:s.offset + s.numel]) /
                                max(np.linalg.norm(refg[s.offset:s.offset + s.numel]), 1e-12) for s in segs
                                if np.linalg.norm(refg[s.offset:s.offset + s.numel]) > 1e-9)
                else:
                    norms = np.array([np.linalg.norm(ga[s.offset:s.offset + s.numel]) for s in segs])
                    rn = g[key + "_gradnorms"]
                    worst = float(np.max(np.abs(norms - rn) / np.maximum(rn, 1e-12)))      # (norm error only: no full gradient stored)
            print(f"{key:20s} {op:5s} {d.max():11.3e} {str(ok):>6s} {abs(loss.item() - float(g[key + '_loss'])):9.2e} {worst:17.3e}",
                  flush=True)


if __name__ == "__main__":
    main()
