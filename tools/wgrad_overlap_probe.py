#!/usr/bin/env python3
"""Would the grouped weight-gradient launch of layer i pay on a side stream, under the dgrad chain of layer i-1?
(The chain of a layer is a dependence chain; the wgrad depends on the layer above only, so it is the one launch of the
backward that can run beside its successors.)  A four-GEMM chain of the step's own kernel families (GEMM+LayerNorm, wide
GELU GEMM, K = 1536 GEMM+LayerNorm, QKV) at M = 32768 and the grouped wgrad of the same layer dimensions:
  (a) one stream: chain, wgrad, chain, wgrad ...
  (b) two free-running streams (upper bound of what concurrency can give)
  (c) two streams with the step's dependences: wgrad(i) waits for chain(i); chain(i+1) waits for wgrad(i-1) (its
      operands double-buffered)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402
from one_wgrad import make_group  # noqa: E402
import two_stream_probe as T  # noqa: E402

lib = _lib.load()


def main():
    M = int(os.environ.get("QST_M", "32768"))
    d = T.mk(M)
    grp, keep, flops = make_group(M=M)
    s0 = torch.cuda.current_stream()
    s1 = torch.cuda.Stream()
    reps = 24

    def serial():
        for _ in range(reps):
            T.chain(d, M, s0.cuda_stream)
            _lib.check(lib.qst_gemm_tn_group(grp, s0.cuda_stream))

    def chain_only():
        for _ in range(reps):
            T.chain(d, M, s0.cuda_stream)

    def wgrad_only():
        for _ in range(reps):
            _lib.check(lib.qst_gemm_tn_group(grp, s0.cuda_stream))

    def free():
        s1.wait_stream(s0)
        for _ in range(reps):
            T.chain(d, M, s0.cuda_stream)
        for _ in range(reps):
            _lib.check(lib.qst_gemm_tn_group(grp, s1.cuda_stream))
        s0.wait_stream(s1)

    evc = [torch.cuda.Event() for _ in range(reps)]
    evw = [torch.cuda.Event() for _ in range(reps)]

    def dep():
        s1.wait_stream(s0)
        for i in range(reps):
            if i >= 2:
                s0.wait_event(evw[i - 2])
            T.chain(d, M, s0.cuda_stream)
            evc[i].record(s0)
            s1.wait_event(evc[i])
            _lib.check(lib.qst_gemm_tn_group(grp, s1.cuda_stream))
            evw[i].record(s1)
        s0.wait_stream(s1)

    res = {}
    for rnd in range(3):
        for name, fn in (("chain only", chain_only), ("wgrad only", wgrad_only), ("(a) one stream", serial),
                         ("(b) two free streams", free), ("(c) two streams, step's dependences", dep)):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / reps * 1e3
            res[name] = min(res.get(name, 1e9), t)
    for k, v in res.items():
        print(f"{k:40s} {v:8.1f} us per layer")


if __name__ == "__main__":
    main()
