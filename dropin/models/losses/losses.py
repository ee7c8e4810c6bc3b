"""Drop-in for /root/reference/models/losses/losses.py: same names, HIP arithmetic."""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)

from quadruplet_sentence_transformer_amd.losses import (DEFAULT_GAMMA, REDUCTIONS, GammaQuadrupletLoss,  # noqa: E402,F401
                                                        QuadrupletLoss, gamma_quadruplet_loss)
