"""Same-process A/B of qst_gemm8_mode on the whole training step.

    python tools/ab_gemm8.py [model] [batch] [seq_len] [rounds] [modes, comma separated: 0 = tiled kernels, 1 = NT on the
                              8-phase path, 2 = weight gradients on it, 3 = both, -1 = the library's own choice] [iters] [bf16 | fp8]
"""
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
    model = sys.argv[1] if len(sys.argv) > 1 else "all-MiniLM-L6-v2"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    modes = [int(x) for x in (sys.argv[5] if len(sys.argv) > 5 else "0,1,2,3").split(",")]
    iters = int(sys.argv[6]) if len(sys.argv) > 6 else 10
    precision = sys.argv[7] if len(sys.argv) > 7 else "bf16"          # "fp8": the fp8-forward training step
    lib = _lib.load()
    cfg = PRESETS[model]
    tr = QuadrupletTrainer(cfg, arena=synthetic_params(cfg, seed=14), device="cuda:0", lr=2e-5, weight_decay=0.01, max_grad_norm=1.0,
                           precision=precision)
    batch = [torch.from_numpy(x).cuda() for x in synthetic_quadruplets(cfg, B, L, seed=14)]
    res = {m: [[], []] for m in modes}
    for _ in range(rounds):
        for m in modes:
            lib.qst_gemm8_mode(m)
            res[m][0].append(timed(lambda: tr.step(*batch), iters))
            res[m][1].append(timed(lambda: tr.forward_loss(*batch, precision=precision), iters))
    lib.qst_gemm8_mode(-1)
    for m in modes:
        st, fw = res[m]
        print(f"gemm8_mode {m:2d}: step {min(st):.3f} ms (runs {' '.join(f'{x:.3f}' for x in st)})   "
              f"forward-only {min(fw):.3f} ms ({B / min(st) * 1e3:.0f} q/s)")


if __name__ == "__main__":
    main()
