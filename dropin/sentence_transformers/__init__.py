"""Drop-in `sentence_transformers` namespace for the reference's unchanged scripts: put `<repo>/dropin` (and the
repo root) on PYTHONPATH ahead of site-packages and `from sentence_transformers import SentenceTransformer,
InputExample` resolves to the MI355X-native implementation."""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)

from quadruplet_sentence_transformer_amd.sentence_transformer import InputExample, SentenceTransformer  # noqa: E402,F401
from . import util, evaluation  # noqa: E402,F401

__version__ = "2.2.2+qst_amd"
