import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd
from quadruplet_sentence_transformer_amd import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace("libqst.so", "libqst_stamp.so")
lib = _lib.load(); st = _lib.current_stream_ptr(); dev="cuda"; bf=torch.bfloat16
M=32768
for name,N,K,epi,force in [("FFN1 fwd epi2 256-row",1536,384,2,2),("FFN1 fwd epi2 128-row",1536,384,2,1),("out fwd epi1 128-row",384,384,1,1),("FFN2 fwd epi1 128-row",384,1536,1,1),("FFN2 dgrad epi3 256-row",1536,384,3,2),("QKV fwd epi0 256-row",1152,384,0,2)]:
    A=torch.randn(M,K,device=dev).to(bf); B=(torch.randn(N,K,device=dev)*0.02).to(bf); bias=torch.zeros(N,device=dev)
    resid=torch.randn(M,N,device=dev); C=torch.empty(M,N,device=dev,dtype=torch.float32 if epi==1 else bf); C2=torch.empty(M,N,device=dev,dtype=bf); aux=torch.randn(M,N,device=dev).to(bf)
    dbg = torch.zeros(8192*8*8, dtype=torch.int64, device=dev)
    g=_lib.QstGemmArgs(); g.A,g.B,g.C,g.C2,g.bias,g.resid,g.aux,g.colsum=A.data_ptr(),B.data_ptr(),C.data_ptr(),C2.data_ptr(),bias.data_ptr(),resid.data_ptr(),aux.data_ptr(),dbg.data_ptr()
    g.M,g.N,g.K,g.lda,g.ldb,g.ldc,g.ldr,g.splits=M,N,K,K,K,N,N,force
    for _ in range(3): _lib.check(lib.qst_gemm_nt(g,epi,st))
    torch.cuda.synchronize()
    rows = 256 if force==2 else 128
    nblk = ((M+rows-1)//rows)*((N+191)//192); nw = 8 if force==2 else 4
    t = dbg.cpu().numpy().reshape(-1, 8).astype(np.float64)
    t = t.reshape(8192, 8, 8)[:nblk, :nw].reshape(-1, 8)
    med = np.median(t, axis=0)
    print(f"{name:26s} prologue {med[0]:6.0f} wait {med[1]:6.0f} barrier {med[2]:6.0f} compute {med[3]:6.0f} epilogue {med[4]:6.0f} [loads-issue {med[6]:5.0f} lds-write {med[7]:5.0f} rest {med[4]-med[6]-med[7]:6.0f}] total {med[5]:6.0f}  (stages={K//64}, MFMA/wave={K//64*24})")
