"""A few launches of the tiled grouped wgrad of one MiniLM layer (M = 32768) with the soft lockstep off / on, for rocprofv3
counter passes (tools/pmc_quick.sh): python tools/one_tnsync.py [0|1]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ab_tnsync import one_layer_group  # noqa: E402
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402

if __name__ == "__main__":
    sync = int(sys.argv[1]) if len(sys.argv) > 1 else int(os.environ.get("QST_TNSYNC", "0"))
    us, n, _ = one_layer_group(_lib.load(), 32768, 384, 1536, sync)
    print(f"sync {sync}: {us:.1f} us, counters {n}")
