import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quadruplet_sentence_transformer_amd
from quadruplet_sentence_transformer_amd import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace("libqst.so", "libqst_stamp.so")
from tools.one_wgrad import make_group
lib = _lib.load(); st = _lib.current_stream_ptr()
grp, keep, flops = make_group()
dbg = torch.zeros(4096*4*8, dtype=torch.int64, device="cuda")
grp.prob[7].A = dbg.data_ptr()
for _ in range(3): _lib.check(lib.qst_gemm_tn_group(grp, st))
torch.cuda.synchronize()
t = dbg.cpu().numpy().reshape(-1, 8).astype(np.float64)
t = t[t[:,5] > 0]
print("waves", len(t), "stages", np.median(t[:,0]))
med = np.median(t, axis=0); mx = np.max(t, axis=0)
print(f"median: wait {med[1]:.0f} barrier {med[2]:.0f} compute {med[3]:.0f} atomics-epilogue {med[4]:.0f} total {med[5]:.0f}")
print(f"max:    wait {mx[1]:.0f} barrier {mx[2]:.0f} compute {mx[3]:.0f} atomics-epilogue {mx[4]:.0f} total {mx[5]:.0f}")
print("per stage: wait %.0f barrier %.0f compute %.0f (18-21 MFMA = 576-672 own cycles)" % (med[1]/med[0], med[2]/med[0], med[3]/med[0]))
