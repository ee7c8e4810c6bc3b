import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd
from quadruplet_sentence_transformer_amd import _lib
from tools.gemm_bench import timeit
lib = _lib.load(); st = _lib.current_stream_ptr(); dev="cuda"; bf=torch.bfloat16
M=32768
for name,N,K,epi in [("FFN1 fwd epi2",1536,384,2),("FFN2 fwd epi1",384,1536,1),("QKV fwd epi0",1152,384,0)]:
    A=torch.randn(M,K,device=dev).to(bf); B=(torch.randn(N,K,device=dev)*0.02).to(bf); bias=torch.zeros(N,device=dev)
    resid=torch.randn(M,N,device=dev); C=torch.empty(M,N,device=dev,dtype=torch.float32 if epi==1 else bf); C2=torch.empty(M,N,device=dev,dtype=bf)
    for label, lda, ldb in (("normal", K, K), ("A rows aliased (lda=8)", 8, K), ("no memory (ld=0 -> OOB zero fill)", 0, 0)):
        g=_lib.QstGemmArgs(); g.A,g.B,g.C,g.C2,g.bias,g.resid=A.data_ptr(),B.data_ptr(),C.data_ptr(),C2.data_ptr(),bias.data_ptr(),resid.data_ptr()
        g.M,g.N,g.K,g.lda,g.ldb,g.ldc,g.ldr,g.splits=M,N,K,lda,ldb,N,N,0
        us=timeit(lambda: _lib.check(lib.qst_gemm_nt(g,epi,st)))
        print(f"{name} {label}: {us:.1f} us")
