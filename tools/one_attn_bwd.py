#!/usr/bin/env python3
"""Time attention backward at a step's shape: the one-workgroup-per-(sequence, head) kernel vs the two-kernel path, results
compared.   python tools/one_attn_bwd.py [nseq L A d] [rel] [drop]      (defaults: 256 128 12 32; mpnet 128 256 12 64 1 1;
bert-base configs[4] 512 384 12 64 0 1)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    n, L, A, d = [int(x) for x in sys.argv[1:5]] if len(sys.argv) > 4 else (256, 128, 12, 32)
    use_rel = len(sys.argv) > 5 and sys.argv[5] == "1"
    use_drop = len(sys.argv) > 6 and sys.argv[6] == "1"
    H = A * d
    lib = _lib.load()
    st = _lib.current_stream_ptr()
    bf = torch.bfloat16
    qkv = torch.randn(n * L, 3 * H, device="cuda").to(bf)
    mask = torch.ones(n, L, dtype=torch.int64, device="cuda")
    ctx = torch.empty(n * L, H, dtype=bf, device="cuda")
    lse = torch.empty(n, A, L, device="cuda")
    rel = (0.5 * torch.randn(A, 2 * L, device="cuda")) if use_rel else None
    state = torch.tensor([14, 0, 3, 0], dtype=torch.int32, device="cuda")
    _lib.check(lib.qst_attention_fwd(qkv.data_ptr(), mask.data_ptr(), _lib.ptr(rel), n, L, A, d, ctx.data_ptr(), lse.data_ptr(), st))
    dctx = torch.randn(n * L, H, device="cuda").to(bf)
    dq = torch.empty(n * L, 3 * H, dtype=bf, device="cuda")
    delta = torch.empty(n, A, L, device="cuda")

    dq2 = torch.empty_like(dq)
    drel = [torch.zeros(A, 2 * L, device="cuda") if use_rel else None for _ in range(2)]

    def desc(split, out, dr):
        q = _lib.QstAttnDesc()
        q.qkv, q.mask, q.rel_pos, q.nseq, q.L, q.A, q.d = qkv.data_ptr(), mask.data_ptr(), _lib.ptr(rel), n, L, A, d
        q.ctx, q.lse, q.dctx, q.dqkv, q.delta_scratch = ctx.data_ptr(), lse.data_ptr(), dctx.data_ptr(), out.data_ptr(), delta.data_ptr()
        q.drel = _lib.ptr(dr)
        q.force_split = split
        if use_drop:
            q.drop.state, q.drop.site, q.drop.thr16 = state.data_ptr(), 2, 6554
        return q
    q1, q2 = desc(2 if d == 64 else 0, dq, drel[0]), desc(1, dq2, drel[1])
    _lib.check(lib.qst_attention_bwd_ex(q1, st))
    _lib.check(lib.qst_attention_bwd_ex(q2, st))
    torch.cuda.synchronize()
    dmax = (dq.float() - dq2.float()).abs().max().item()
    l2 = ((dq.float() - dq2.float()).norm() / dq2.float().norm()).item()
    extra = ""
    if use_rel:
        extra = f", drel rel L2 {((drel[0] - drel[1]).norm() / drel[1].norm()).item():.2e}"
    best = [1e9, 1e9]
    for _ in range(4):
        best[0] = min(best[0], timeit(lambda: _lib.check(lib.qst_attention_bwd_ex(q1, st))))
        best[1] = min(best[1], timeit(lambda: _lib.check(lib.qst_attention_bwd_ex(q2, st))))
    print(f"attention backward n={n} L={L} A={A} d={d} rel={int(use_rel)} drop={int(use_drop)}: one workgroup per (sequence, head) "
          f"{best[0]:.1f} us, two-kernel {best[1]:.1f} us; max|d| {dmax:.3g}, rel L2 {l2:.2e}{extra}")


if __name__ == "__main__":
    main()
