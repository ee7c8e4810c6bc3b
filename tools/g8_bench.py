"""8-phase GEMM path (csrc/gemm8.hip) against the tiled kernels of csrc/gemm.hip: results and alternating best-of timings.

    python tools/g8_bench.py [H] [M]         H = 384 (MiniLM) or 768 (mpnet / bert-base); M token rows (default 32768)

NT: every epilogue on the layer's shapes, forms: 0 = gemm.hip's own choice, 0x20 = 8-phase 128 x 384, 0x40 = 8-phase 256 x 256.
TN: the layer's four weight gradients in one grouped launch, qst_gemm8_mode 0 / 2.
Timings are back-to-back launches (operands warm); the in-step figures come from bench.py / rocprofv3.
"""
import math
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402

lib = _lib.load()
H = int(sys.argv[1]) if len(sys.argv) > 1 else 768
M = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
I = 4 * H
st = _lib.current_stream_ptr


def args(**kw):
    g = _lib.QstGemmArgs()
    g._keep = [v for v in kw.values() if torch.is_tensor(v)]
    for k, v in kw.items():
        setattr(g, k, v.data_ptr() if torch.is_tensor(v) else v)
    return g


def timeit(fns, rounds=6, iters=10):
    """alternating best-of: every candidate is timed in every round"""
    best = [1e9] * len(fns)
    for _ in range(rounds):
        for i, f in enumerate(fns):
            f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            best[i] = min(best[i], e0.elapsed_time(e1) / iters * 1e3)
    return best


g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)

print(f"# H = {H}, M = {M}")
print("NT  N     K     epi            old us   8p128x384   8p256x256   (TF/s of the best 8p)   max|d| vs old")
EPI = {0: "bf16", 1: "f32+resid", 2: "gelu", 3: "gelu_bwd", 4: "f32+resid+bf16"}
shapes = [(3 * H, H, 0), (H, H, 1), (H, H, 4), (I, H, 2), (H, I, 4), (I, H, 3), (H, I, 1), (H, 3 * H, 1), (H, H, 0)]
for N, K, epi in shapes:
    A = (rnd(M, K)).to(torch.bfloat16)
    B = (rnd(N, K) * 0.05).to(torch.bfloat16)
    bias = rnd(N)
    resid = rnd(M, N)
    aux = (torch.rand(M, N, device="cuda", generator=g) * 1.2 - 0.1).to(torch.bfloat16)
    outs = {}

    def run(form):
        f32 = epi in (1, 4)
        C = torch.empty(M, N, dtype=torch.float32 if f32 else torch.bfloat16, device="cuda")
        C2 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        a = args(A=A, B=B, C=C, C2=C2, aux=aux, bias=bias, resid=resid, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, ldr=N, splits=form)
        outs[form] = (C, C2)
        return lambda: _lib.check(lib.qst_gemm_nt(a, epi, st()))

    fns = [run(0x80), run(0x20), run(0x40)]
    t = timeit(fns)
    d1 = (outs[0x20][0].float() - outs[0x80][0].float()).abs().max().item()
    d2 = (outs[0x40][0].float() - outs[0x80][0].float()).abs().max().item()
    if epi in (2, 4):
        d1 = max(d1, (outs[0x20][1].float() - outs[0x80][1].float()).abs().max().item())
        d2 = max(d2, (outs[0x40][1].float() - outs[0x80][1].float()).abs().max().item())
    fl = 2.0 * M * N * K
    print(f"    {N:5d} {K:5d} {EPI[epi]:14s} {t[0]:7.1f}  {t[1]:9.1f}  {t[2]:9.1f}   {fl / min(t[1:]) * 1e-6:8.0f}            {d1:.3g} {d2:.3g}")

print("TN  grouped wgrad of one layer")
probs = [(3 * H, H), (H, H), (I, H), (H, I)]
grp = _lib.QstTnGroup()
grp.nprob, grp.splits = 4, 0
keep, Cs = [], []
for i, (N, K) in enumerate(probs):
    A = rnd(M, N).to(torch.bfloat16)
    B = rnd(M, K).to(torch.bfloat16)
    C = torch.zeros(N, K, device="cuda")
    cs = torch.zeros(N, device="cuda")
    q = grp.prob[i]
    q.A, q.B, q.C, q.colsum = A.data_ptr(), B.data_ptr(), C.data_ptr(), cs.data_ptr()
    q.M, q.N, q.K, q.lda, q.ldb, q.ldc = M, N, K, N, K, K
    keep += [A, B]
    Cs.append((C, cs, A, B))


def tn(mode, splits=0):
    def f():
        lib.qst_gemm8_mode(mode)
        grp.splits = splits
        _lib.check(lib.qst_gemm_tn_group(grp, st()))
        grp.splits = 0
        lib.qst_gemm8_mode(-1)
    return f


res = {}
for mode in (0, 2):
    for C, cs, _, _ in Cs:
        C.zero_(); cs.zero_()
    tn(mode)()
    torch.cuda.synchronize()
    res[mode] = [(C.clone(), cs.clone()) for C, cs, _, _ in Cs]
for i, (C, cs, A, B) in enumerate(Cs):
    ref = A.float().t() @ B.float()
    e_old = (res[0][i][0] - ref).abs().max().item()
    e_new = (res[2][i][0] - ref).abs().max().item()
    b_new = (res[2][i][1] - A.float().sum(0)).abs().max().item()
    print(f"    problem {i} {tuple(ref.shape)}: max|d| vs fp32 matmul old {e_old:.3g} new {e_new:.3g} (scale {math.sqrt(M):.0f}); bias grad new {b_new:.3g}")
cands = [tn(0), tn(2)] + ([tn(2, -256)] if H % 256 == 0 else [])
t = timeit(cands)
fl = 2.0 * M * sum(n * k for n, k in probs)
print(f"    tiled {t[0]:.1f} us ({fl / t[0] * 1e-6:.0f} TF/s)   8-phase 128 x 384 {t[1]:.1f} us ({fl / t[1] * 1e-6:.0f} TF/s)"
      + (f"   8-phase 256 x 256 {t[2]:.1f} us" if len(t) > 2 else ""))
