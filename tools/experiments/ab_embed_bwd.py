#!/usr/bin/env python3
"""qst_embed_bwd (one float atomic per element of every token row) against qst_embed_bwd_sorted (word rows grouped by id
first) at the step's shapes: alternating best-of timing on ds rows that are cold (600 MB of unrelated writes between
launches would cost too much here: the two arms simply alternate), results compared.

    python tools/ab_embed_bwd.py [nseq L H vocab] [zipf]      # zipf: ids drawn from a 1/rank distribution (real text)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402


def main():
    a = sys.argv[1:]
    nseq, L, H, V = (int(x) for x in a[:4]) if len(a) >= 4 else (256, 128, 384, 30522)
    zipf = "zipf" in a
    lib = _lib.load()
    M = nseq * L
    g = torch.Generator().manual_seed(1)
    if zipf:
        w = 1.0 / torch.arange(1, V + 1, dtype=torch.float64)
        ids = torch.multinomial(w, M, replacement=True, generator=g)
    else:
        ids = torch.randint(0, V, (M,), generator=g)
    ds = torch.randn(M, H, generator=g).cuda()
    ids = ids.cuda()
    types = torch.zeros(M, dtype=torch.int64, device="cuda")
    pos = torch.arange(L, dtype=torch.int32).repeat(nseq).cuda()
    order = torch.empty(M, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def tables():
        return (torch.zeros(V, H, device="cuda"), torch.zeros(512, H, device="cuda"), torch.zeros(2, H, device="cuda"))

    def old(t):
        _lib.check(lib.qst_embed_bwd(ds.data_ptr(), ids.data_ptr(), types.data_ptr(), pos.data_ptr(), nseq, L, H, 2,
                                     t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), st))

    def new(t):
        _lib.check(lib.qst_embed_bwd_sorted(ds.data_ptr(), ids.data_ptr(), types.data_ptr(), pos.data_ptr(), nseq, L, H, 2, V,
                                            t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), order.data_ptr(), st))

    ta, tb = tables(), tables()
    old(ta); new(tb)
    torch.cuda.synchronize()
    for x, y, nm in zip(ta, tb, ("word", "pos", "type")):
        print(f"{nm}: max |old - sorted| = {(x - y).abs().max().item():.3e} (max |old| {x.abs().max().item():.2f})")
    best = {"atomics": 1e9, "sorted": 1e9}
    t = tables()
    for rnd in range(6):
        for nm, fn in (("atomics", old), ("sorted", new)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                fn(t)
            e1.record()
            torch.cuda.synchronize()
            best[nm] = min(best[nm], e0.elapsed_time(e1) / 10 * 1e3)
    uniq = int(torch.unique(ids).numel())
    print(f"nseq={nseq} L={L} H={H} vocab={V} {'zipf' if zipf else 'uniform'} ids ({uniq} distinct of {M}): "
          f"atomics {best['atomics']:.1f} us, sorted {best['sorted']:.1f} us")


if __name__ == "__main__":
    main()
