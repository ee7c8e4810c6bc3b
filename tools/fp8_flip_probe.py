"""the fp8 TRAINING forward of tools/fuzz_shapes.py seed 53 case 26 (mpnet dims, 2 layers, 5 quadruplets, L = 32,
dropout 0.2 / 0.05 -- embeddings 4.37e-3 from the MX oracle against the suite's 4e-3) against the oracle's TWO accumulation
orders: fp32 through the host BLAS (what the suite runs) and fp64. Both oracles quantise exactly as the kernels do; they differ
from each other, and from the HIP path, only in the last bits of the fp32 value that enters the next quantiser -- and an e4m3
rounding that flips moves its element by 2^-3 relative. Prints each side's distance to the other two.
usage: python tools/fp8_flip_probe.py [more (family layers B L p_hidden p_attn) ...]"""
import io
import os
import re
import sys
from contextlib import redirect_stdout
from dataclasses import replace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from quadruplet_sentence_transformer_amd.config import PRESETS  # noqa: E402
from oracle import torch_ref as R  # noqa: E402
import test_gpu_fp8mx as T8  # noqa: E402


def run(fam, layers, B, L, drop):
    PRESETS["fuzz"] = replace(PRESETS[fam], num_layers=layers, vocab_size=2048)
    out = {}
    for acc64 in (False, True):
        R.MX_ACC64 = acc64
        buf = io.StringIO()
        status = "ok"
        try:
            with redirect_stdout(buf):
                T8.test_fp8_training_step_against_the_mx_oracle("fuzz", B, L, layers, dict(std=0.03, bias_std=0.02, ln_jitter=0.05), drop)
        except AssertionError as ex:
            status = "AssertionError " + str(ex).split("\n")[0][:80]
        m = re.search(r"embeddings ([0-9.e+-]+), loss ([0-9.e+-]+)", buf.getvalue())
        g = re.search(r"grad-cls\] .*?: (.*)", buf.getvalue())
        out[acc64] = (m.group(1) if m else "?", m.group(2) if m else "?", g.group(1) if g else "", status)
    R.MX_ACC64 = False
    print(f"{fam} layers={layers} B={B} L={L} dropout={drop}")
    for acc64 in (False, True):
        e, l, gr, st = out[acc64]
        print(f"   HIP vs oracle accumulating in {'fp64' if acc64 else 'fp32 (BLAS)'}: embeddings {e} (relative to the mean norm), loss {l}; {gr}  [{st}]")


if __name__ == "__main__":
    run("all-mpnet-base-v2", 2, 5, 32, (0.2, 0.05))
    run("all-mpnet-base-v2", 2, 5, 32, None)
    run("all-MiniLM-L6-v2", 2, 4, 64, (0.1, 0.1))
    run("bert-base-uncased", 1, 2, 96, (0.1, 0.1))
