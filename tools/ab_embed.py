#!/usr/bin/env python3
"""qst_embed_bwd of several builds of libqst.so in ONE process (alternating, best of 5 x 20 launches), at the step's shapes, with
uniformly random and Zipf-distributed word ids; results compared against the first library's.

    python tools/ab_embed.py tools/libqst_base.so tools/libqst_x.so ..."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402


def bind(path):
    lib = C.CDLL(os.path.abspath(path))
    res, args = _lib.SIGNATURES["qst_embed_bwd"]
    lib.qst_embed_bwd.restype, lib.qst_embed_bwd.argtypes = res, args
    return lib


def main():
    _lib.load()
    libs = [(p, bind(p)) for p in sys.argv[1:]]
    st = torch.cuda.current_stream().cuda_stream
    for nseq, L, H, V, zipf in ((256, 128, 384, 30522, False), (256, 128, 384, 30522, True), (128, 256, 768, 30527, False),
                                (512, 384, 768, 30522, False)):
        M = nseq * L
        g = torch.Generator().manual_seed(1)
        if zipf:
            ids = torch.multinomial(1.0 / torch.arange(1, V + 1, dtype=torch.float64), M, replacement=True, generator=g)
        else:
            ids = torch.randint(0, V, (M,), generator=g)
        ds = torch.randn(M, H, generator=g).cuda()
        ids = ids.cuda()
        types = torch.zeros(M, dtype=torch.int64, device="cuda")
        pos = torch.arange(L, dtype=torch.int32).repeat(nseq).cuda()
        tabs = [(torch.zeros(V, H, device="cuda"), torch.zeros(512, H, device="cuda"), torch.zeros(2, H, device="cuda")) for _ in libs]

        def run(i):
            t = tabs[i]
            _lib.check(libs[i][1].qst_embed_bwd(ds.data_ptr(), ids.data_ptr(), types.data_ptr(), pos.data_ptr(), nseq, L, H, 2,
                                                t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), st))
        for i in range(len(libs)):
            run(i)
        torch.cuda.synchronize()
        errs = [max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(tabs[i], tabs[0])) for i in range(len(libs))]
        best = [1e9] * len(libs)
        for _ in range(5):
            for i in range(len(libs)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                run(i)
                e0.record()
                for _ in range(20):
                    run(i)
                e1.record()
                torch.cuda.synchronize()
                best[i] = min(best[i], e0.elapsed_time(e1) / 20 * 1e3)
        print(f"nseq={nseq} L={L} H={H} {'zipf' if zipf else 'uniform'} ids: " +
              "  ".join(f"{os.path.basename(p)} {b:.1f} us (rel diff {e:.1e})" for (p, _), b, e in zip(libs, best, errs)))


if __name__ == "__main__":
    main()
