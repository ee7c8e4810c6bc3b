"""The 128 x 192 NT kernel with its activation stages through registers two ahead (QstGemmArgs.splits bit 8; qst_gemm8_mode
bit 2) against the LDS-DMA form: every NT launch of a MiniLM layer back to back (alternating, best of), outputs compared, and the
whole training step with the mode alternated.   python tools/ab_aregs.py [model] [B] [L] [rounds]"""
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


def gemm_args(**kw):
    g = _lib.QstGemmArgs()
    g._keep = [v for v in kw.values() if torch.is_tensor(v)]
    for k, v in kw.items():
        setattr(g, k, v.data_ptr() if torch.is_tensor(v) else v)
    return g


def bench(lib, M, N, K, epi, name):
    gen = torch.Generator().manual_seed(M + N + K + epi)
    A = (torch.randn(M, K, generator=gen)).to(torch.bfloat16).cuda()
    B = (torch.randn(N, K, generator=gen) * 0.05).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=gen).cuda()
    resid = torch.randn(M, N, generator=gen).cuda() if epi in (1, 4) else None
    aux = (torch.rand(M, N, generator=gen)).to(torch.bfloat16).cuda() if epi == 3 else None
    outs, times = {}, {1: [], 0x101: []}
    flush = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    for rep in range(4):
        for form in (1, 0x101):
            C = torch.empty(M, N, dtype=torch.float32 if epi in (1, 4) else torch.bfloat16, device="cuda")
            C2 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            g = gemm_args(A=A, B=B, C=C, C2=C2, bias=bias if epi != 3 else None, resid=resid, aux=aux, M=M, N=N, K=K, lda=K, ldb=K, ldc=N,
                          ldr=N, splits=form)
            st = _lib.current_stream_ptr()
            lib.qst_gemm_nt(g, epi, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            tot = 0.0
            for _ in range(5):
                flush.fill_(1)                                   # operands cold, as inside the step (DESIGN.md finding 19)
                e0.record()
                lib.qst_gemm_nt(g, epi, st)
                e1.record()
                torch.cuda.synchronize()
                tot += e0.elapsed_time(e1)
            times[form].append(tot / 5 * 1e3)
            outs[form] = (C.float().clone(), C2.float().clone())
    same = torch.equal(outs[1][0], outs[0x101][0]) and (epi != 2 or torch.equal(outs[1][1], outs[0x101][1]))
    print(f"{name:28s} M={M} N={N} K={K}: LDS-DMA {min(times[1]):7.1f} us   A through registers {min(times[0x101]):7.1f} us   "
          f"bit-identical {same}", flush=True)


def timed(fn, iters):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "all-MiniLM-L6-v2"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    lib = _lib.load()
    cfg = PRESETS[model]
    M, H, I = 4 * B * L, cfg.hidden_size, cfg.intermediate_size
    bench(lib, M, 3 * H, H, 0, "QKV (bf16 out)")
    bench(lib, M, I, H, 2, "FFN-1 + GELU")
    bench(lib, M, I, H, 3, "GELU' dgrad")
    bench(lib, M, H, H, 0, "out-proj dgrad (bf16 out)")
    bench(lib, M, H, H, 1, "out-proj + residual (fp32)")
    bench(lib, M, H, I, 1, "FFN-2 + residual (fp32)")
    tr = QuadrupletTrainer(cfg, arena=synthetic_params(cfg, seed=14), device="cuda:0", lr=2e-5, weight_decay=0.01, max_grad_norm=1.0,
                           dropout=0.1, dropout_seed=14)
    batches = [[torch.from_numpy(x).cuda() for x in synthetic_quadruplets(cfg, B, L, seed=14, step=i)] for i in range(4)]
    import ctypes as C
    lib.qst_gemm_nt_aregs.restype, lib.qst_gemm_nt_aregs.argtypes = C.c_int, [C.c_int]
    mink = int(sys.argv[5]) if len(sys.argv) > 5 else 64
    res = {0: [], mink: []}
    for _ in range(rounds):
        for m in (0, mink):
            lib.qst_gemm_nt_aregs(m)
            res[m].append(timed(lambda i=0: tr.step(*batches[i % 4]), 10 if M > 40000 else 20))
    lib.qst_gemm_nt_aregs(0)
    for m in (0, mink):
        st = res[m]
        print(f"A through registers for K >= {m} (0 = off): step best {min(st):.3f} ms, mean "
              f"{sum(st) / len(st):.3f} (runs {' '.join(f'{x:.3f}' for x in st)})  {B / min(st) * 1e3:.0f} q/s")


if __name__ == "__main__":
    main()
