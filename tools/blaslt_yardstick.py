#!/usr/bin/env python3
"""Yardstick only (not on the product path): what the vendor library reaches on the step's GEMM shapes, plain bf16
C = A.B^T with no epilogue, so the hand-written kernels' main loops can be judged against something measured."""
import torch


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
    M = 32768
    for H in (384, 768):
        I = 4 * H
        for name, N, K in [("QKV", 3 * H, H), ("out", H, H), ("FFN1", I, H), ("FFN2", H, I)]:
            A = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
            B = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
            C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            us = timeit(lambda: torch.matmul(A, B.t(), out=C))
            print(f"H={H} {name:5s} NT M={M} N={N} K={K}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")
        for name, N, K in [("dW1", I, H), ("dW2", H, I)]:
            dY = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
            X = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
            us = timeit(lambda: torch.matmul(dY.t(), X))
            print(f"H={H} {name:5s} TN M={M} N={N} K={K}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")
    # a large square GEMM: what fraction of 2.5 PF the library itself reaches
    A = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    B = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    us = timeit(lambda: torch.matmul(A, B.t()))
    print(f"8192^3 NT: {us:7.1f} us  {2.0 * 8192 ** 3 / us / 1e6:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
