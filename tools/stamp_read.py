import os, sys, torch, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd
from quadruplet_sentence_transformer_amd import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace("libqst.so", "libqst_stamp.so")
lib = _lib.load(); st = _lib.current_stream_ptr(); dev="cuda"; bf=torch.bfloat16
M=32768; N=1152; K=384
A=torch.randn(M,K,device=dev).to(bf); B=(torch.randn(N,K,device=dev)*0.02).to(bf); bias=torch.zeros(N,device=dev)
C=torch.empty(M,N,device=dev,dtype=bf)
nblk = (M//128)*(N//128)
dbg = torch.zeros(nblk*4*24, dtype=torch.int64, device=dev)
g=_lib.QstGemmArgs(); g.A,g.B,g.C,g.bias,g.aux=A.data_ptr(),B.data_ptr(),C.data_ptr(),bias.data_ptr(),dbg.data_ptr()
g.M,g.N,g.K,g.lda,g.ldb,g.ldc=M,N,K,K,K,N
for _ in range(3): _lib.check(lib.qst_gemm_nt(g,0,st))
torch.cuda.synchronize()
t = dbg.cpu().numpy().reshape(nblk*4, 24).astype(np.float64)
d = np.diff(t[:, :22], axis=1)
names = ["prologue issue"] + sum([[f"s{k} wait vmcnt", f"s{k} barrier", f"s{k} compute+issue"] for k in range(6)], []) + ["final barrier", "epilogue"]
idx = list(range(0,1)) + list(range(1,19)) + [19, 20]
med = np.median(d, axis=0)
print("total median cycles per wave:", np.median(t[:,21]-t[:,0]))
for n_, i in zip(names, idx): print(f"{n_:24s} {med[i]:9.0f}")
