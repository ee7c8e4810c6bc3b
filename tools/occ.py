import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "quadruplet-sentence-transformer_amd", "libqst.so"))
for which, name, lds in [(0, "fwd", 2*8192+512), (1, "dq", 3*8192+512), (2, "dkv", 4*8192+1024)]:
    print(name, "lds", lds, "-> blocks/CU", lib.qst_debug_attn_occupancy(which, lds), "| with 8KB:", lib.qst_debug_attn_occupancy(which, 8192))
