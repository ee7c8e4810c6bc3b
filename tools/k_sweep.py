import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd
from quadruplet_sentence_transformer_amd import _lib
from tools.gemm_bench import timeit
lib = _lib.load(); st = _lib.current_stream_ptr(); dev="cuda"; bf=torch.bfloat16
M=32768
for epi, N in [(0, 1536), (2, 1536), (1, 384), (0, 384)]:
    for K in (64, 128, 384, 768, 1536):
        A=torch.randn(M,K,device=dev).to(bf); B=(torch.randn(N,K,device=dev)*0.02).to(bf); bias=torch.zeros(N,device=dev)
        resid=torch.randn(M,N,device=dev); C=torch.empty(M,N,device=dev,dtype=torch.float32 if epi==1 else bf); C2=torch.empty(M,N,device=dev,dtype=bf)
        g=_lib.QstGemmArgs(); g.A,g.B,g.C,g.C2,g.bias,g.resid=A.data_ptr(),B.data_ptr(),C.data_ptr(),C2.data_ptr(),bias.data_ptr(),resid.data_ptr()
        g.M,g.N,g.K,g.lda,g.ldb,g.ldc,g.ldr=M,N,K,K,K,N,N
        us=timeit(lambda: _lib.check(lib.qst_gemm_nt(g,epi,st)))
        out_mb = M*N*(4 if epi==1 else 2)*(2 if epi in (1,2) else 1)/1e6 + M*K*2/1e6
        print(f"epi{epi} N={N} K={K:5d}: {us:7.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF  ({out_mb:.0f} MB -> {out_mb/us*1e-3*1e3:.2f} TB/s)")
