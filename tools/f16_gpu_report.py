"""GPU: per golden case, the HIP forward / backward at precision bf16, f16 and bf16x3 against the committed HF vectors --
max |d emb|, elements outside rtol 1e-3 / atol 1e-4, |d loss|, worst per-tensor gradient error (relative L2 for the cases
that store full gradients, relative norm error otherwise) -- and, for the small cases, the f16 path against the f16-operand
oracle (oracle/torch_ref.py bf16_operands="f16"). Usage: python tools/f16_gpu_report.py [--oracle]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd.config import PRESETS, build_layout  # noqa: E402
from quadruplet_sentence_transformer_amd.encoder import HipEncoder, quadruplet_loss_raw, stacked  # noqa: E402
from quadruplet_sentence_transformer_amd.synthetic import synthetic_params  # noqa: E402
from tests.test_oracle_golden import CLI, ENC_CASES, golden_inputs  # noqa: E402

CASES = ENC_CASES + [("minilm_l128", "all-MiniLM-L6-v2", 2, 128, dict(std=0.02), "norms")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--oracle", action="store_true")
    a = ap.parse_args()
    g = np.load(os.path.join(ROOT, "tests", "golden", "encoder_golden.npz"))
    print(f"{'case':20s} {'prec':7s} {'max|d emb|':>11s} {'outside':>12s} {'|d loss|':>9s} {'worst grad':>11s} {'which':20s}")
    for key, preset, B, L, wkw, store in CASES:
        cfg = PRESETS[preset]
        arena = synthetic_params(cfg, seed=14, **wkw)
        ids, mask, types = golden_inputs(key, cfg, B, L)
        segs, total = build_layout(cfg)
        n = 4 * B
        dev = [torch.from_numpy(x).view(n, L).cuda() for x in (ids, mask, types)]
        if not cfg.type_vocab_size:
            dev[2] = None
        ref = g[key + "_emb"]
        for prec in ("bf16", "f16", "f16w", "bf16x3"):
            enc = HipEncoder(cfg)
            enc.load_arena(arena)
            emb, _, saved = enc.forward(*dev, training=True, precision=prec)
            e4 = emb.view(4, B, -1)
            S = 65536.0 if prec in ("f16", "f16w") else 1.0
            gout = torch.tensor([S], dtype=torch.float32, device="cuda")
            loss, gr = quadruplet_loss_raw(e4[0], e4[1], e4[2], e4[3], 0.6, 1.0, 0.5, 0.5, 2.0, False, 2, grad_out=gout, want_grads=True)
            enc.ensure_train_state()
            enc.grads.zero_()
            enc.backward(*dev, stacked(gr), saved, precision=prec)
            ga = enc.grads.cpu().numpy() / S
            d = np.abs(e4.cpu().numpy() - ref)
            bad = int((d > 1e-4 + 1e-3 * np.abs(ref)).sum())
            worst, which = 0.0, ""
            if store == "full":
                rg = g[key + "_grads"]
                for s in segs:
                    b = rg[s.offset:s.offset + s.numel]
                    if np.linalg.norm(b) < 1e-9:
                        continue
                    e = np.linalg.norm(ga[s.offset:s.offset + s.numel] - b) / np.linalg.norm(b)
                    if e > worst:
                        worst, which = e, s.name
            else:
                rn = g[key + "_gradnorms"]
                for k, s in enumerate(segs):
                    if rn[k] < 1e-9:
                        continue
                    e = abs(np.linalg.norm(ga[s.offset:s.offset + s.numel]) - rn[k]) / rn[k]
                    if e > worst:
                        worst, which = e, s.name + " (norm)"
            print(f"{key:20s} {prec:7s} {d.max():11.3e} {bad:6d}/{d.size:<6d} {abs(loss.item() - float(g[key + '_loss'])):9.2e} "
                  f"{worst:11.3e} {which:20s}", flush=True)
            if a.oracle and prec == "f16" and B * L * cfg.num_layers <= 2048 * 6:
                from oracle import torch_ref as R
                P = R.arena_to_dict(arena, cfg, requires_grad=True)
                lo, eo = R.quadruplet_step(P, cfg, torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(types), CLI,
                                           bf16_operands="f16")
                (lo * S).backward()
                w2, wh2 = 0.0, ""
                for s in segs:
                    b = P[s.name].grad.numpy().reshape(-1) / S
                    if np.linalg.norm(b) < 1e-9:
                        continue
                    e = np.linalg.norm(ga[s.offset:s.offset + s.numel] - b) / np.linalg.norm(b)
                    if e > w2:
                        w2, wh2 = e, s.name
                print(f"{'':20s} {'vs f16 oracle':13s} max|d emb| {np.abs(e4.cpu().numpy() - eo.detach().numpy()).max():.3e}  "
                      f"|d loss| {abs(loss.item() - lo.item()):.2e}  worst grad {w2:.3e} {wh2}", flush=True)
            del enc


if __name__ == "__main__":
    main()
