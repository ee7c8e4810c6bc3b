import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd
from quadruplet_sentence_transformer_amd import _lib
lib = _lib.load(); st = _lib.current_stream_ptr(); dev="cuda"; bf=torch.bfloat16
M=32768; N=1536; K=384; epi=int(sys.argv[1]) if len(sys.argv)>1 else 2
A=torch.randn(M,K,device=dev).to(bf); B=(torch.randn(N,K,device=dev)*0.02).to(bf); bias=torch.zeros(N,device=dev)
resid=torch.randn(M,N,device=dev); C=torch.empty(M,N,device=dev,dtype=torch.float32 if epi==1 else bf); C2=torch.empty(M,N,device=dev,dtype=bf)
aux=torch.randn(M,N,device=dev).to(bf)
g=_lib.QstGemmArgs(); g.A,g.B,g.C,g.C2,g.bias,g.resid,g.aux=A.data_ptr(),B.data_ptr(),C.data_ptr(),C2.data_ptr(),bias.data_ptr(),resid.data_ptr(),aux.data_ptr()
g.M,g.N,g.K,g.lda,g.ldb,g.ldc,g.ldr,g.splits=M,N,K,K,K,N,N,0
for _ in range(5): _lib.check(lib.qst_gemm_nt(g,epi,st))
torch.cuda.synchronize()
