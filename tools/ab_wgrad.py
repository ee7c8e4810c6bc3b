#!/usr/bin/env python3
"""A/B two builds of libqst.so on the grouped wgrad launch (MiniLM and mpnet layer shapes) in ONE process.
usage: ab_wgrad.py old.so [new.so]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402
from one_wgrad import make_group  # noqa: E402


def bind(path):
    lib = C.CDLL(path)
    res, args = _lib.SIGNATURES["qst_gemm_tn_group"]
    lib.qst_gemm_tn_group.restype, lib.qst_gemm_tn_group.argtypes = res, args
    return lib


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
    libs = [bind(os.path.abspath(sys.argv[1])), bind(os.path.abspath(sys.argv[2]) if len(sys.argv) > 2 else _lib.LIB_PATH)]
    st = _lib.current_stream_ptr()
    for H in (384, 768):
        grp, keep, flops = make_group(H=H, I=4 * H)
        best = [1e9, 1e9]
        for _ in range(3):
            for i, lib in enumerate(libs):
                best[i] = min(best[i], timeit(lambda: _lib.check(lib.qst_gemm_tn_group(grp, st))))
        print(f"H={H}: old {best[0]:7.1f} us ({flops / best[0] / 1e6:6.1f} TF)   new {best[1]:7.1f} us ({flops / best[1] / 1e6:6.1f} TF)   ({best[1] / best[0] - 1:+.1%})")


if __name__ == "__main__":
    main()
