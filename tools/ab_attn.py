#!/usr/bin/env python3
"""Same-process A/B of the attention kernels of two builds of the library at the step's shape (256 sequences x 128
tokens, 12 heads of 32): tools/libqst_base.so (build it from the commit to compare with: `git archive <rev>
quadruplet-sentence-transformer_amd/csrc include | tar -x -C /tmp/base && make -C /tmp/base/.../csrc`) against the
in-tree libqst.so. Forward and backward, with and without dropout of the probabilities, alternating, best of 5 rounds;
also checks that the two builds agree (bit for bit unless the arithmetic changed: the max difference is printed).

    python tools/ab_attn.py [nseq] [L] [A] [d]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    A = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    d = int(sys.argv[4]) if len(sys.argv) > 4 else 32
    H = A * d
    libs = {"base": C.CDLL(os.path.join(ROOT, "tools", "libqst_base.so")), "new": _lib.load()}
    for lb in libs.values():
        for f in ("qst_attention_fwd_ex", "qst_attention_bwd_ex"):
            getattr(lb, f).argtypes = [C.POINTER(_lib.QstAttnDesc), C.c_void_p]
            getattr(lb, f).restype = C.c_int
        lb.qst_dropout_init.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    st = _lib.current_stream_ptr()
    bf = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(5)
    qkv = torch.randn(n * L, 3 * H, device="cuda", generator=g).to(bf)
    lens = torch.randint(L // 4, L + 1, (n,), device="cuda", generator=g)
    mask = (torch.arange(L, device="cuda")[None, :] < lens[:, None]).long().contiguous()
    dctx = torch.randn(n * L, H, device="cuda", generator=g).to(bf)
    state = torch.zeros(4, dtype=torch.int32, device="cuda")
    libs["new"].qst_dropout_init(state.data_ptr(), 1234, st)
    res = {}
    out = {}
    for drop in (False, True):
        bufs = {}
        for name, lb in libs.items():
            ctx = torch.empty(n * L, H, dtype=bf, device="cuda")
            lse = torch.empty(n, A, L, device="cuda")
            dq = torch.empty(n * L, 3 * H, dtype=bf, device="cuda")
            delta = torch.empty(n, A, L, device="cuda")
            q = _lib.QstAttnDesc()
            q.qkv, q.mask, q.nseq, q.L, q.A, q.d = qkv.data_ptr(), mask.data_ptr(), n, L, A, d
            q.ctx, q.lse, q.dctx, q.dqkv, q.delta_scratch = ctx.data_ptr(), lse.data_ptr(), dctx.data_ptr(), dq.data_ptr(), delta.data_ptr()
            if drop:
                q.drop.state, q.drop.site, q.drop.thr16 = state.data_ptr(), 3, int(0.1 * 65536)
            bufs[name] = (q, ctx, lse, dq, delta)
        for rnd in range(5):
            for name, lb in libs.items():
                q = bufs[name][0]
                tf = timeit(lambda: _lib.check(lb.qst_attention_fwd_ex(q, st)))
                tb = timeit(lambda: _lib.check(lb.qst_attention_bwd_ex(q, st)))
                k = (name, drop)
                res[k] = (min(res.get(k, (1e9, 1e9))[0], tf), min(res.get(k, (1e9, 1e9))[1], tb))
        torch.cuda.synchronize()
        d_ctx = (bufs["base"][1].float() - bufs["new"][1].float()).abs().max().item()
        d_dq = (bufs["base"][3].float() - bufs["new"][3].float()).abs().max().item()
        ref = bufs["base"][3].float().abs().max().item()
        out[drop] = (d_ctx, d_dq, ref)
    for drop in (False, True):
        b, nw = res[("base", drop)], res[("new", drop)]
        print(f"n={n} L={L} A={A} d={d} dropout={'on ' if drop else 'off'}: forward base {b[0]:6.1f} us new {nw[0]:6.1f} us | "
              f"backward base {b[1]:6.1f} us new {nw[1]:6.1f} us | max |ctx diff| {out[drop][0]:.3g}, "
              f"max |dqkv diff| {out[drop][1]:.3g} (max |dqkv| {out[drop][2]:.3g})")


if __name__ == "__main__":
    main()
