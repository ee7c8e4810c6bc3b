#!/usr/bin/env python3
"""Diagnostic: build libqst with -DQST_STAMP_ATTN (tools/libqst_stamp.so), run the single-workgroup attention
backward at the step's shape and print the cycle split of one item per phase (median over workgroups).
  python tools/stamp_attn_bwd.py build   (here, cross-compiles)
  python tools/stamp_attn_bwd.py         (on the GPU box)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "quadruplet-sentence-transformer_amd", "csrc")
SO = os.path.join(ROOT, "tools", "libqst_stamp.so")

if len(sys.argv) > 1 and sys.argv[1] == "build":
    flags = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -DQST_STAMP_ATTN".split()
    subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-c", os.path.join(SRC, "attention.hip"), "-o", "/tmp/attention_stamp.o"])
    open("/tmp/stamp_stub.cpp", "w").write('extern "C" void qst_set_hip_error(int) {}\n')
    subprocess.check_call(["g++", "-fPIC", "-c", "/tmp/stamp_stub.cpp", "-o", "/tmp/stamp_stub.o"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, "/tmp/attention_stamp.o",
                           "/tmp/stamp_stub.o"])
    print("built", SO)
    sys.exit(0)

import numpy as np
import torch

lib = C.CDLL(SO)
vp = C.c_void_p
lib.qst_attention_fwd.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
lib.qst_attention_bwd.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
n, L, A, d = 256, 128, 12, 32
H = A * d
bf = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
qkv = torch.randn(n * L, 3 * H, device="cuda").to(bf)
mask = torch.ones(n, L, dtype=torch.int64, device="cuda")
ctx = torch.empty(n * L, H, dtype=bf, device="cuda")
lse = torch.empty(n, A, L, device="cuda")
assert lib.qst_attention_fwd(qkv.data_ptr(), mask.data_ptr(), None, n, L, A, d, ctx.data_ptr(), lse.data_ptr(), st) == 0
dctx = torch.randn(n * L, H, device="cuda").to(bf)
dq = torch.empty(n * L, 3 * H, dtype=bf, device="cuda")
stamps = torch.zeros(512 * 16, dtype=torch.int64, device="cuda")
for _ in range(3):
    assert lib.qst_attention_bwd(qkv.data_ptr(), ctx.data_ptr(), dctx.data_ptr(), lse.data_ptr(), mask.data_ptr(), None,
                                 n, L, A, d, dq.data_ptr(), None, stamps.data_ptr(), st) == 0
torch.cuda.synchronize()
t = stamps.cpu().numpy().reshape(512, 16).astype(np.int64)
names = ["stage regs->LDS + delta (+ issue next prefetch)", "wait barrier", "4 score tiles (S, dP, exp, dV, dK, dS image)",
         "wait barrier", "dQ from the dS image", "stores"]
print("cycles per phase, median over 512 workgroups (third item of each):")
for k, nm in enumerate(names):
    dlt = t[:, k + 1] - t[:, k]
    print(f"  {nm:50s} {np.median(dlt):9.0f}   (p10 {np.percentile(dlt, 10):.0f}, p90 {np.percentile(dlt, 90):.0f})")
print(f"  {'  of the first phase: regs->LDS + delta':50s} {np.median(t[:, 7] - t[:, 0]):9.0f}")
print(f"  {'  of the first phase: issuing the next prefetch':50s} {np.median(t[:, 1] - t[:, 7]):9.0f}")
print(f"  {'whole item':50s} {np.median(t[:, 6] - t[:, 0]):9.0f}")
print("inside the second score tile:")
for k, nm in ((8, "issue a quarter of the next item's loads"), (9, "row constants + S / dP MFMAs (issue)"),
              (10, "elementwise: exp, dS, dS image rows"), (11, "transposed reads + dV / dK MFMAs (issue)")):
    dlt = t[:, k + 1] - t[:, k]
    print(f"  {nm:50s} {np.median(dlt):9.0f}   (p10 {np.percentile(dlt, 10):.0f}, p90 {np.percentile(dlt, 90):.0f})")
life = t[:, 14] - t[:, 13]
start = t[:, 13] - t[:, 13].min()
print(f"workgroup life (6 items): median {np.median(life):.0f} cycles, p10 {np.percentile(life, 10):.0f}, p90 {np.percentile(life, 90):.0f}, "
      f"max {life.max():.0f}; start skew median {np.median(start):.0f}, max {start.max():.0f}; "
      f"first start -> last end {t[:, 14].max() - t[:, 13].min():.0f}")
xcd = np.arange(512) % 8
for x in range(8):
    print(f"  XCD {x}: median life {np.median(life[xcd == x]):.0f}, max {life[xcd == x].max():.0f}")
