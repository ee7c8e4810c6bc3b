import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd
from quadruplet_sentence_transformer_amd import _lib
lib = _lib.load(); st = _lib.current_stream_ptr(); dev="cuda"; bf=torch.bfloat16
n, L, A, d = 256, 128, 12, 32
H = A*d
qkv = torch.randn(n*L, 3*H, device=dev).to(bf); mask = torch.ones(n, L, dtype=torch.int64, device=dev)
ctx = torch.empty(n*L, H, device=dev, dtype=bf); lse = torch.empty(n, A, L, device=dev)
delta = torch.empty(n, A, L, device=dev); dctx = torch.randn(n*L, H, device=dev).to(bf); dqkv = torch.empty(n*L, 3*H, device=dev, dtype=bf)
for _ in range(5):
    _lib.check(lib.qst_attention_fwd(qkv.data_ptr(), mask.data_ptr(), None, n, L, A, d, ctx.data_ptr(), lse.data_ptr(), st))
    _lib.check(lib.qst_attention_bwd(qkv.data_ptr(), ctx.data_ptr(), dctx.data_ptr(), lse.data_ptr(), mask.data_ptr(), None, n, L, A, d, dqkv.data_ptr(), None, delta.data_ptr(), st))
torch.cuda.synchronize()
