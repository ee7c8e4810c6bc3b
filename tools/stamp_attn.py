import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd
from quadruplet_sentence_transformer_amd import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace("libqst.so", "libqst_stamp.so")
lib = _lib.load(); st = _lib.current_stream_ptr(); dev="cuda"; bf=torch.bfloat16
n, L, A, d = 256, 128, 12, 32
H = A*d
qkv = torch.randn(n*L, 3*H, device=dev).to(bf); mask = torch.ones(n, L, dtype=torch.int64, device=dev)
ctx = torch.empty(n*L, H, device=dev, dtype=bf); lse = torch.empty(n, A, L, device=dev); delta = torch.empty(n, A, L, device=dev)
dctx = torch.randn(n*L, H, device=dev).to(bf); dqkv = torch.empty(n*L, 3*H, device=dev, dtype=bf)
dbg = torch.zeros(n*A*4*8, dtype=torch.int64, device=dev)
_lib.check(lib.qst_attention_fwd(qkv.data_ptr(), mask.data_ptr(), None, n, L, A, d, ctx.data_ptr(), lse.data_ptr(), st))
for _ in range(3):
    _lib.check(lib.qst_attention_bwd(qkv.data_ptr(), ctx.data_ptr(), dctx.data_ptr(), lse.data_ptr(), mask.data_ptr(), None, n, L, A, d, dqkv.data_ptr(), dbg.data_ptr(), delta.data_ptr(), st))
torch.cuda.synchronize()
t = dbg.cpu().numpy().reshape(-1, 8)[:, :6].astype(np.float64)
dd = np.diff(t, axis=1)
med = np.median(dd, axis=0)
print("dkv per wave (cycles): frag-load issue->stage start %.0f | stage loads until stores %.0f | stores+barrier %.0f | q-tile loop %.0f | output stores %.0f | total %.0f" % (med[0], med[1], med[2], med[3], med[4], np.median(t[:,5]-t[:,0])))
