"""Same-process A/B of the weight-gradient kernel's soft lockstep (QstTnGroup.sync, qst_gemm_tn_sync_mode): the whole training
step with the counters off / on, alternated, cycling four batches with dropout as bench.py does; and the grouped launch of one
layer alone, back to back.

    python tools/ab_tnsync.py [model] [batch] [seq_len] [rounds] [iters]
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
    for i in range(iters):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def one_layer_group(lib, M, H, I, sync):
    g = torch.Generator().manual_seed(1)
    shapes = [(H, I), (I, H), (H, H), (3 * H, H)]
    grp = _lib.QstTnGroup()
    grp.nprob, grp.splits = 4, 0
    keep = []
    for i, (N, K) in enumerate(shapes):
        A = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
        B = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
        C = torch.zeros(N, K, device="cuda")
        cs = torch.zeros(N, device="cuda")
        q = grp.prob[i]
        q.A, q.B, q.C, q.colsum = A.data_ptr(), B.data_ptr(), C.data_ptr(), cs.data_ptr()
        q.M, q.N, q.K, q.lda, q.ldb, q.ldc = M, N, K, N, K, K
        keep += [A, B, C, cs]
    n = lib.qst_gemm_tn_sync_ints(grp)
    buf = torch.zeros(max(n, 1) + 64, dtype=torch.int32, device="cuda")
    if sync and n > 0:
        grp.sync, grp.sync_ints = buf.data_ptr(), n
    st = _lib.current_stream_ptr()
    for _ in range(3):
        _lib.check(lib.qst_gemm_tn_group(grp, st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.qst_gemm_tn_group(grp, st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3, n, keep[2].clone()


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "all-MiniLM-L6-v2"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
    lib = _lib.load()
    cfg = PRESETS[model]
    M = 4 * B * L
    for sync in (0, 1, 0, 1):
        us, n, c = one_layer_group(lib, M, cfg.hidden_size, cfg.intermediate_size, sync)
        print(f"grouped wgrad of one layer, M = {M}: sync {sync}: {us:.1f} us  (counters {n}; |dW2| checksum {float(c.abs().sum()):.6e})", flush=True)
    tr = QuadrupletTrainer(cfg, arena=synthetic_params(cfg, seed=14), device="cuda:0", lr=2e-5, weight_decay=0.01, max_grad_norm=1.0,
                           dropout=0.1, dropout_seed=14)
    batches = [[torch.from_numpy(x).cuda() for x in synthetic_quadruplets(cfg, B, L, seed=14, step=i)] for i in range(4)]
    res = {0: [], 1: []}
    for _ in range(rounds):
        for m in (0, 1):
            lib.qst_gemm_tn_sync_mode(m)
            res[m].append(timed(lambda i=0: tr.step(*batches[i % 4]), iters))
    lib.qst_gemm_tn_sync_mode(1)
    for m in (0, 1):
        st = res[m]
        print(f"tn_sync_mode {m}: step best {min(st):.3f} ms, mean {sum(st) / len(st):.3f} (runs {' '.join(f'{x:.3f}' for x in st)})  "
              f"{B / min(st) * 1e3:.0f} q/s")


if __name__ == "__main__":
    main()
